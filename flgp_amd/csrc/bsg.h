// Block-sparse products with the Gram matrix G = A^T A for the eigensolver's Chebyshev filter (bsg.hip).
//
// G couples two anchors only if some point has both among its r nearest: 5 % of G is non-zero at BASELINE configs[2]
// (s = 5000, r = 10, 238 entries per row).  G is the weight matrix of a neighbourhood graph with cluster / manifold
// structure, so a symmetric permutation concentrates it: after the ordering built here 9 % of the 64 x 16 blocks of
// P G P^T hold 98.8 % of the non-zeros.  A product then is
//     (kept blocks: MFMA GEMM over the listed 16-deep k stages of every 64-row tile)  +  (the scattered rest: CSR),
// on blocks stored transposed (b x s, the b values of one anchor contiguous).  The ordering can only change how much
// work is skipped, never a result beyond rounding.
#pragma once
#include "common.h"
#include <vector>

namespace flgp {

constexpr int BSG_TM = 64;        // anchors per output tile
constexpr int BSG_TN = 64;        // block columns per output tile
constexpr int BSG_SK = 16;        // k depth of a stage
constexpr int BSG_SEEDS = 64;     // clusters of the ordering
constexpr int BSG_META = 16;      // ints of device -> host bookkeeping

struct BsG {
  bool built = false;             // set-up enqueued; `on` is decided by bsg_finish() once the stream has been synchronised
  bool on = false;
  int s = 0, ntile = 0, nstage = 0, nbt = 0;      // nbt: 64-column tiles of the b-wide blocks the workspace was carved for
  // CSR of G in the caller's anchor order
  int *gptr = nullptr, *gcol = nullptr;
  double *gval = nullptr;
  size_t csr_cap = 0;
  // ordering
  int *lab = nullptr, *perm = nullptr, *iperm = nullptr;
  double *E0 = nullptr, *E1 = nullptr, *Cw = nullptr;
  // kept blocks of P G P^T
  int *cnt = nullptr, *blkpos = nullptr, *nk = nullptr, *off = nullptr, *klist = nullptr, *order = nullptr;
  double *pack = nullptr;
  size_t pack_cap = 0;            // blocks
  // A tile's list is cut into parts of at most `bs_part_cap` stages, each part a task of its own (bsg.hip, product kernel):
  // (tile, first list entry, entries, part | parts << 16), longest first; slab0[tile] = the tile's first partial-sum
  // slab, tcnt = arrival counters per (tile, column tile), zero between launches
  int *parts = nullptr, *parts_u = nullptr, *slab0 = nullptr, *tcnt = nullptr;
  double *slab = nullptr;
  size_t part_max = 0, slab_cap = 0;   // parts the lists may hold; parts of split tiles the slab memory holds
  // the scattered rest, rows in permuted order
  int *rcnt = nullptr, *rptr = nullptr, *rcol = nullptr;
  double *rval = nullptr;
  size_t rem_cap = 0;
  // per-row figures of G gathered while the CSR is counted: absolute column sums, diagonal
  double *colabs = nullptr, *diag = nullptr, *bounds = nullptr;
  // Host copies of what the set-up reads back (asynchronously: valid after the next synchronisation of their stream).
  // They point at the struct's own arrays unless the caller hands in pinned slots before bsg_setup (bsg_host_slots): a
  // copy into pageable memory may block the host until the stream reaches it, and one into a stack frame that an early
  // return has left is a write into dead memory -- the eigensolver passes slots its per-thread context owns.
  double h_bounds_own[2] = {0.0, 0.0};
  double *h_bounds = h_bounds_own;   // 1-norm of G (>= lambda_max), trace of G
  int *meta = nullptr;            // device: see BSG_M_* in bsg.hip
  int *head = nullptr;            // work-queue heads of the product kernel (ring of 2)
  double *lz = nullptr;           // Lanczos vectors / partial sums / (alpha, beta)
  double h_ab_own[64] = {0};
  double *h_ab = h_ab_own;        // (alpha_k, beta_k) of the Lanczos steps, valid after lz_ev
  bool lanczos = false;
  hipEvent_t lz_ev = nullptr;
  double lambda_lo = 0.0;         // safe-side estimate of lambda_min(G) (0 when there is none): set by bsg_finish
  double *T[3] = {nullptr, nullptr, nullptr};   // b x s blocks of the transposed filter
  int h_meta_own[BSG_META] = {0};
  int *h_meta = h_meta_own;       // host copy of meta
  int launches = 0;               // products issued
  std::vector<int> h_perm;        // source of an asynchronous upload: lives as long as the struct
  void *h_big = nullptr;          // caller-owned pinned memory for the set-up's larger exchange (bsg_host_slots), or null
  size_t h_big_bytes = 0;
};

constexpr size_t BSG_HOST_SLOT_BYTES = sizeof(double) * (2 + 64) + sizeof(int) * BSG_META;
// point the read-back copies at caller-owned (pinned) host memory of BSG_HOST_SLOT_BYTES bytes; `big` (optional, big_bytes
// of pinned memory) takes the set-up's one larger exchange with the host: cluster weights + labels down, permutation up
void bsg_host_slots(BsG &g, void *slots, void *big = nullptr, size_t big_bytes = 0);
size_t bsg_workspace_bytes(int s, int b);
// carve the members out of a workspace (advances p)
void bsg_carve(BsG &g, char *&p, int s, int b);
// Enqueues the whole set-up (one host round trip in the middle, for the 64-cluster chain).  On return the device work
// is queued and the bookkeeping is on its way to g.h_meta: call bsg_finish() after the next stream synchronisation and
// before the first product.
int bsg_setup(hipStream_t st, const double *dG, int ldg, int s, BsG &g, hipStream_t side = nullptr, hipEvent_t side_ev = nullptr);
// after a synchronisation: decides g.on (false when G does not concentrate: the caller multiplies with the dense G)
void bsg_finish(BsG &g);
// out_t = alpha X_t (P G P^T) + beta E_t + gamma E2_t   (all b x s, b contiguous; E_t / E2_t may be null; out_t must not
// alias X_t)
int bsg_product(hipStream_t st, BsG &g, const double *Xt, int b, double alpha, double beta, const double *Et,
                double gamma, const double *E2t, double *out_t);

}  // namespace flgp
