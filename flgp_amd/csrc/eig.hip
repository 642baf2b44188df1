// k6 (dense half): top-K eigenpairs of the s x s Gram matrix G = A^T A on the device.
//
// Replaces the RSpectra::svds call of truncated_SVD_cpp (reference src/TruncatedSVD.cpp:23-30:
// Spectra's implicitly restarted Lanczos on the implicit s x s operator, one vector at a time,
// entered through an R callback) and its K == s Eigen::BDCSVD branch (:17-20).
//
// MI355X design: a BLOCK method, so that every pass over G is an fp64 MFMA GEMM instead of a
// memory-bound mat-vec -- Chebyshev-filtered subspace iteration:
//     Y  <- p_m(G) Q          m GEMMs (s x s x b) with the three-term recurrence in the epilogue
//     Q1 <- orth(Y)           S = Y^T Y (split-K GEMM), eigen-decomposition of S (block Jacobi),
//                             Q1 = Y W L^-1/2  ("SVQB": no Cholesky breakdown on ill-conditioned Y)
//     Rayleigh-Ritz           Z = G Q1, T = Q1^T Z, T = Yr Th Yr^T (block Jacobi), Q = Q1 Yr, GQ = Z Yr
//     residuals               |G q_j - th_j q_j| for the K wanted pairs -> host decides to stop
// p_m damps [0, th_b] (G is PSD) and is scaled to 1 at the top Ritz value; b = K + guard columns.
// A block of width b >= the multiplicity of lambda = 1 (one per connected component of the
// anchor graph) resolves repeated eigenvalues, which single-vector Lanczos only finds through
// round-off.
//
// The small dense symmetric eigenproblems (b x b; or s x s itself when K == s / s is small)
// use a one-sided block Jacobi: block-column pairs are orthogonalised inside LDS by one
// workgroup each, rounds follow a round-robin tournament, one launch per round (independent
// pairs in a round; a launch boundary is the inter-workgroup barrier).  Rotations are
// accumulated in V explicitly, so V stays orthogonal to rounding even for tiny eigenvalues.
#include "common.h"
#include "bsg.h"
#include "host_wait.h"
#include <algorithm>
#include <mutex>
#include <cmath>
#include <vector>

namespace flgp {

// Host wait for the stream.  The solver talks to the host ~50 times per solve (Jacobi convergence flags, Ritz
// values, the Newton-Schulz checks); hipStreamSynchronize sleeps on an interrupt and costs ~35 us of idle GPU per
// round trip, polling an event costs a few.
// The HIP objects a solve needs on the host side -- a second stream and three events -- come from a small pool
// (creating and destroying a stream costs the better part of a millisecond; callers may solve from short-lived
// threads, e.g. the bandwidth grid, so thread-local objects would pile up).  A solve borrows one set for its
// duration; the pool only ever grows to the number of solves that were in flight at once.
struct HostCtx {
  int device = -1;
  hipStream_t side = nullptr;
  hipEvent_t side_ev = nullptr, mark_ev = nullptr, wait_ev = nullptr;
  void *pinned = nullptr;      // page-locked slots for the solve's small asynchronous read-backs (HOST_SLOT_BYTES)
  void *pinned_dev = nullptr;  // the same memory as the device addresses it (null: kernels cannot write it, copies are used)
};
constexpr int DIST_SLOT_DOUBLES = 64;          // partial sums dist_to_identity_kernel writes straight into host memory
constexpr int RT_SLOT_DOUBLES = 2 * 4096;      // residuals + sorted Ritz values of a Rayleigh-Ritz step (2 b doubles), likewise
constexpr int APRIORI_SLOT_DOUBLES = 512;      // 2 x APRIORI_BLOCKS (checked where the blocks are defined)
constexpr size_t HOST_SMALL_BYTES = BSG_HOST_SLOT_BYTES + sizeof(double) * (APRIORI_SLOT_DOUBLES + DIST_SLOT_DOUBLES + RT_SLOT_DOUBLES);
constexpr size_t HOST_BIG_BYTES = 512 << 10;   // the block-sparse set-up's exchange: 32 KB of cluster weights + 8 bytes per anchor
constexpr size_t HOST_SLOT_BYTES = HOST_SMALL_BYTES + HOST_BIG_BYTES;
static std::mutex g_ctx_mu;
static std::vector<HostCtx *> g_ctx_free;
static thread_local HostCtx *g_ctx = nullptr;     // the set borrowed by the solve running on this thread

struct HostCtxLease {
  HostCtx *prev;
  HostCtxLease() : prev(g_ctx) {
    HostCtx *c = nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      for (size_t q = 0; q < g_ctx_free.size(); ++q)
        if (g_ctx_free[q]->device == dev) { c = g_ctx_free[q]; g_ctx_free.erase(g_ctx_free.begin() + q); break; }
    }
    if (!c) {
      c = new HostCtx();
      c->device = dev;
      if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
      if (hipEventCreateWithFlags(&c->side_ev, hipEventDisableTiming) != hipSuccess) c->side_ev = nullptr;
      if (hipEventCreateWithFlags(&c->mark_ev, hipEventDisableTiming) != hipSuccess) c->mark_ev = nullptr;
      if (hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming) != hipSuccess) c->wait_ev = nullptr;
      // coherent (fine-grained) and mapped: kernels write a step's verdict straight into it, the host reads it behind the
      // event that follows the kernel -- no copy launch, no staging (a D2H copy of 256 bytes into pageable memory was a blit
      // kernel plus a host memcpy per question, ~45 questions per solve)
      if (hipHostMalloc(&c->pinned, HOST_SLOT_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        if (hipHostMalloc(&c->pinned, HOST_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pinned = nullptr; }
      }
      if (c->pinned && hipHostGetDevicePointer(&c->pinned_dev, c->pinned, 0) != hipSuccess) { (void)hipGetLastError(); c->pinned_dev = nullptr; }
    }
    g_ctx = c;
  }
  ~HostCtxLease() {
    {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      g_ctx_free.push_back(g_ctx);
    }
    g_ctx = prev;
  }
};

// stream_mark: remember this point of the stream; mark_wait: host waits until the stream has reached it.
// Both waits poll the event (a round trip costs a few us instead of the ~35 us of an interrupt), but only for
// `eig_spin_us` microseconds (default 2000: the longest stretch between two questions of a healthy solve is a
// Rayleigh-Ritz step of ~1.2 ms); after that the thread blocks in hipEventSynchronize like any other HIP caller, so
// a kernel that never finishes costs an idle thread, not a spinning one, and a device error surfaces as FLGP_ERR_HIP.
static hipError_t event_wait(hipEvent_t ev) {
  const long spin_us = tuning("eig_spin_us", 2000);
  const int rc = bounded_wait(
      [&]() -> int { const hipError_t e = hipEventQuery(ev); return e == hipSuccess ? 0 : (e == hipErrorNotReady ? 1 : 1000 + (int)e); },
      [&]() -> int { const hipError_t e = hipEventSynchronize(ev); return e == hipSuccess ? 0 : 1000 + (int)e; }, spin_us);
  return rc == 0 ? hipSuccess : (hipError_t)(rc - 1000);
}
static hipError_t stream_mark(hipStream_t st) {
  if (!g_ctx || !g_ctx->mark_ev) return hipStreamSynchronize(st);
  return hipEventRecord(g_ctx->mark_ev, st);
}
static hipError_t mark_wait() {
  if (!g_ctx || !g_ctx->mark_ev) return hipSuccess;    // stream_mark synchronised instead
  return event_wait(g_ctx->mark_ev);
}
hipError_t stream_wait(hipStream_t st) {
  if (!g_ctx || !g_ctx->wait_ev || tuning("eig_spin_wait", 1) == 0) return hipStreamSynchronize(st);
  const hipError_t e = hipEventRecord(g_ctx->wait_ev, st);
  if (e != hipSuccess) return e;
  return event_wait(g_ctx->wait_ev);
}


// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// deterministic start block: uniform(-1,1) from a counter hash
__global__ void eig_init_q_kernel(double *__restrict__ Q, int s, int b, int ldq, unsigned long long stream) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * b) return;
  const int i = (int)(e % s), j = (int)(e / s);
  const unsigned long long h = mix64((unsigned long long)e * 2 + 0x5851F42D4C957F2Dull + stream * 0x9E3779B97F4A7C15ull);
  Q[(size_t)j * ldq + i] = ((double)(h >> 11) + 0.5) * (2.0 / 9007199254740992.0) - 1.0;
}

// out = a * X + b * Y (elementwise, s x b, same ld)
__global__ void eig_axpby_kernel(double a, const double *__restrict__ X, double b, const double *__restrict__ Y,
                                 double *__restrict__ out, long total) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) out[e] = a * X[e] + b * Y[e];
}

// Jacobi set-up: B = T (symmetrised), V = I
__global__ void jac_init_kernel(const double *__restrict__ T, int ldt, int b, double *__restrict__ B,
                                double *__restrict__ V, int ldb, int *__restrict__ flags) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) { flags[0] = 0; flags[1] = 0; }
  if (e >= (long)b * b) return;
  const int i = (int)(e % b), j = (int)(e / b);
  B[(size_t)j * ldb + i] = 0.5 * (T[(size_t)j * ldt + i] + T[(size_t)i * ldt + j]);
  V[(size_t)j * ldb + i] = (i == j) ? 1.0 : 0.0;
}

// round-robin tournament pairing of m2 (even) players, round rr in [0, m2-1), pair k in [0, m2/2)
__device__ __host__ inline void rr_pair(int m2, int rr, int k, int &p, int &q) {
  const int m = m2 - 1;
  if (k == 0) { p = m; q = rr % m; }
  else { p = (rr + k) % m; q = (rr + m - k) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

// 1/sqrt(x) to double precision: hardware estimate + two Newton steps (x > 0, normal range)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * (y * y));
  y = y * (1.5 - (0.5 * x) * (y * y));
  return y;
}

// One workgroup orthogonalises the 2w columns of block columns (I, J) against each other:
// one cyclic sweep of Hestenes rotations inside LDS, applied to B and accumulated into V.
// flags[0]: number of rotations applied in this sweep (convergence when it stays 0)
// flags[1]: "converged" latch set by the host-side protocol (kernel exits early)
__global__ __launch_bounds__(1024) void jac_round_kernel(double *__restrict__ B, double *__restrict__ V, int b,
                                                        int ldb, int w, int nbc, int round, double tol,
                                                        int *__restrict__ flags, int local_sweeps) {
  extern __shared__ double sm[];
  __shared__ int any_rot;
  if (flags[1]) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  int I, J;
  rr_pair(nbc, round, blockIdx.x, I, J);
  const int cI = I * w, cJ = J * w;
  if (cI >= b) return;                      // padding block column
  const int wI = (b - cI < w) ? b - cI : w;
  const int wJ = (cJ >= b) ? 0 : ((b - cJ < w) ? b - cJ : w);
  const int ncol = wI + wJ;                 // live columns
  const int bp = b + 16;                    // padded column stride in LDS
  double *LB = sm;                          // [2w][bp]
  double *LV = sm + (size_t)2 * w * bp;
  // load
  for (int c = 0; c < ncol; ++c) {
    const int gc = (c < wI) ? cI + c : cJ + (c - wI);
    for (int i = tid; i < b; i += nt) {
      LB[(size_t)c * bp + i] = B[(size_t)gc * ldb + i];
      LV[(size_t)c * bp + i] = V[(size_t)gc * ldb + i];
    }
  }
  __syncthreads();
  if (ncol >= 2) {
    const int m2 = (ncol + 1) & ~1;         // players (one bye when ncol is odd)
    const int npair = m2 / 2;
    // threads per pair: a power of two <= 64 so that the reductions stay inside a wave
    int gs = 64;
    while (gs > 1 && gs * npair > nt) gs >>= 1;
    const int grp = tid / gs, gl = tid % gs;
    int rotations = 0;
    for (int ls = 0; ls < local_sweeps; ++ls) {
    if (tid == 0) any_rot = 0;
    __syncthreads();
    int rot_here = 0;
    for (int rr = 0; rr < m2 - 1; ++rr) {
      for (int k0 = 0; k0 < npair; k0 += nt / gs) {   // more pairs than groups: several passes
        const int k = k0 + grp;
        int p = 0, q = 0;
        bool act = k < npair;
        if (act) { rr_pair(m2, rr, k, p, q); act = (q < ncol); }
        double al = 0.0, be = 0.0, ga = 0.0;
        constexpr int RC = 8;                 // rows of the pair kept in registers between dots and rotation
        double xr[RC], yr[RC];
        double *bp_ = LB + (size_t)p * bp, *bq_ = LB + (size_t)q * bp;
        if (act) {
#pragma unroll
          for (int c = 0; c < RC; ++c) {
            const int i = gl + c * gs;
            xr[c] = (i < b) ? bp_[i] : 0.0;
            yr[c] = (i < b) ? bq_[i] : 0.0;
            al = __builtin_fma(xr[c], xr[c], al); be = __builtin_fma(yr[c], yr[c], be); ga = __builtin_fma(xr[c], yr[c], ga);
          }
          for (int i = gl + RC * gs; i < b; i += gs) {
            const double x = bp_[i], y = bq_[i];
            al = __builtin_fma(x, x, al); be = __builtin_fma(y, y, be); ga = __builtin_fma(x, y, ga);
          }
        }
        for (int off = gs >> 1; off > 0; off >>= 1) {
          al += __shfl_xor(al, off, 64); be += __shfl_xor(be, off, 64); ga += __shfl_xor(ga, off, 64);
        }
        // |ga| > tol sqrt(al be), squared to stay off the sqrt unit
        if (act && ga * ga > (tol * tol) * (al * be) && al > 0.0 && be > 0.0) {
          // rotation with tan(2 th) = 2 ga / (be - al), |th| <= pi/4, from two reciprocal square roots
          // (v_rsq_f64 + two Newton steps each) instead of three IEEE divisions and two square roots:
          //   r1 = 1/sqrt(dl^2 + 4 ga^2), cos(2th) = |dl| r1, x = (1 + cos 2th)/2, r2 = 1/sqrt(x),
          //   c = x r2, s = sign(dl) ga r1 r2        (c^2 + s^2 = 1 to rounding)
          const double dl = be - al;
          const double r1 = rsqrt_nr(dl * dl + 4.0 * (ga * ga));
          const double x2 = 0.5 + 0.5 * (__builtin_fabs(dl) * r1);
          const double r2 = rsqrt_nr(x2);
          const double cs = x2 * r2;
          const double sn = (dl >= 0.0 ? ga : -ga) * (r1 * r2);
          double *vp_ = LV + (size_t)p * bp, *vq_ = LV + (size_t)q * bp;
#pragma unroll
          for (int c = 0; c < RC; ++c) {
            const int i = gl + c * gs;
            if (i < b) {
              bp_[i] = cs * xr[c] - sn * yr[c]; bq_[i] = sn * xr[c] + cs * yr[c];
              const double u = vp_[i], v = vq_[i];
              vp_[i] = cs * u - sn * v; vq_[i] = sn * u + cs * v;
            }
          }
          for (int i = gl + RC * gs; i < b; i += gs) {
            const double x = bp_[i], y = bq_[i];
            bp_[i] = cs * x - sn * y; bq_[i] = sn * x + cs * y;
            const double u = vp_[i], v = vq_[i];
            vp_[i] = cs * u - sn * v; vq_[i] = sn * u + cs * v;
          }
          if (gl == 0) ++rot_here;
        }
      }
      __syncthreads();
    }
    rotations += rot_here;
    if (rot_here) any_rot = 1;
    __syncthreads();
    if (!any_rot) break;           // this block of columns is orthogonal to working precision
    __syncthreads();
    }
    if (rotations) atomicAdd(&flags[0], rotations);
  }
  // store
  for (int c = 0; c < ncol; ++c) {
    const int gc = (c < wI) ? cI + c : cJ + (c - wI);
    for (int i = tid; i < b; i += nt) {
      B[(size_t)gc * ldb + i] = LB[(size_t)c * bp + i];
      V[(size_t)gc * ldb + i] = LV[(size_t)c * bp + i];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Block-Jacobi visit with the heavy parts on the matrix cores (b % 16 == 0, NLOC = 2w local columns).
//
// One workgroup (16 waves) makes the NLOC columns of block columns (I, J) mutually orthogonal:
//   1. the B panel (b x NLOC) is staged in LDS, column-major with column stride b+2
//      (the 16-column MFMA fragment reads then hit 16 distinct bank pairs);
//   2. Gram block Gm = P^T P (NLOC x NLOC) by v_mfma_f64_16x16x4_f64, the b rows split over 4 waves
//      per 16 x 16 tile and summed in a fixed order;
//   3. two-sided cyclic Jacobi on the small Gm inside LDS, accumulating the rotations in Wm
//      (rotation angles from Gm_pp, Gm_qq, Gm_pq: exactly the one-sided Hestenes angles);
//   4. P <- P Wm by MFMA (each wave owns 16 panel rows: reads them all, then overwrites them),
//      stored back; the V panel goes through the same LDS buffer and the same Wm.
// Per visit the panels are written to LDS twice instead of once per local round (LDS stores are
// the slow direction on CDNA4), which is what made the scalar version (jac_round_kernel) slow.
// ------------------------------------------------------------------------------------------
typedef double jd4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

// diagnostic (flgp_dev_jac_set_trace): word 0 = launches recorded so far, then 8 words per launch of jac_block_kernel,
// written by workgroup 0: 100 MHz wall clock at start / panel in LDS / Gram block summed / small problem solved /
// B panel stored / end, cross_only, round
static long long *g_jac_trace = nullptr;
__device__ __forceinline__ void jac_stamp(long long *tr, int slot) {
  if (tr && blockIdx.x == 0 && threadIdx.x == 0) tr[slot] = (long long)wall_clock64();
}

template <int NLOC>
__global__ __launch_bounds__(1024) void jac_block_kernel(double *__restrict__ B, double *__restrict__ V, int b,
                                                         int ldb, int nbc, int round, double tol,
                                                         int *__restrict__ flags, int local_sweeps,
                                                         int cross_only, long long *__restrict__ trace, int batched_load) {
  constexpr int WB = NLOC / 2;        // block-column width
  constexpr int NP = NLOC / 2;        // pairs per local round
  constexpr int NT16 = NLOC / 16;     // 16-wide tiles per side
  constexpr int TILES = NT16 * NT16;
  constexpr int KP = 4;               // row parts of the Gram product
  extern __shared__ double sm[];
  __shared__ int any_rot, visit_rot;
  __shared__ int round_rot[3];
  if (flags[1]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long *tr = nullptr;
  if (trace && blockIdx.x == 0) {       // launches of one solve follow each other on one stream: the count is stable while this one runs
    const long long n = trace[0];
    tr = (n < 4096) ? trace + 1 + 8 * n : nullptr;
  }
  jac_stamp(tr, 0);
  if (tid < 3) round_rot[tid] = 0;
  if (tid == 0) visit_rot = 0;
  int I, J;
  rr_pair(nbc, round, blockIdx.x, I, J);
  const int cI = I * WB, cJ = J * WB;
  if (cI >= b) return;
  const int bp = b + 2;
  double *P = sm;                         // [NLOC][bp]   column-major panel
  double *Gm = P + (size_t)NLOC * bp;     // [NLOC][NLOC]
  double *Wm = Gm + NLOC * NLOC;          // [NLOC][NLOC] row-major: Wm[k][c]
  double *part = Wm + NLOC * NLOC;        // [KP][NLOC][NLOC]
  auto gcol = [&](int c) { return (c < WB) ? cI + c : cJ + (c - WB); };

  // the way in: two rows per lane (16 bytes), half the vector-memory and LDS-write instructions (b, ldb and bp are even, the
  // matrices 16-byte aligned): 3.3 -> 2.9 us.  (The way out is apply_w's own.)
  // (round 4: as a loop this was load -> s_waitcnt vmcnt(0) -> ds_write per 16 bytes, four memory latencies in a row -- the
  //  panels were last written by other XCDs -- 3.4 of a visit's 26 us; now all of a wave's loads are issued, then stored)
  auto load_panel = [&](const double *M) {
    constexpr int CPW = NLOC / 16;      // columns per wave
    if (b <= 512 && batched_load) {
      d2v tmp[CPW][4];
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int gc = gcol(wave + 16 * cc);
        const double *src = M + (size_t)(gc < b ? gc : 0) * ldb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 2 * lane + 128 * q;
          tmp[cc][q] = *(const d2v *)(src + (i < b ? i : b - 2));       // unconditional: a valid address, masked below
        }
      }
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int c = wave + 16 * cc;
        const bool live = gcol(c) < b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 2 * lane + 128 * q;
          if (i < b) *(d2v *)(P + (size_t)c * bp + i) = live ? tmp[cc][q] : d2v{0.0, 0.0};
        }
      }
      return;
    }
    for (int c = wave; c < NLOC; c += 16) {
      const int gc = gcol(c);
      for (int i = 2 * lane; i < b; i += 128)
        *(d2v *)(P + (size_t)c * bp + i) = (gc < b) ? *(const d2v *)(M + (size_t)gc * ldb + i) : d2v{0.0, 0.0};
    }
  };
  // M(:, the pair's columns) <- P Wm : each wave owns row blocks of 16 rows and stores its results straight to global memory
  // -- D(row = local column fk + 4 reg, col = panel row fr): the 16 lanes of one fk write 128 contiguous bytes of a column
  // of M.  (Through LDS and store_panel, as until round 3, the way out cost two more barriers and 16 LDS instructions per
  // thread and panel.)
  constexpr int VRB = 2;             // row blocks per wave whose V operands are held in registers (b <= 512)
  auto apply_w = [&](double *M, const double (*pre)[NLOC / 4]) {
    int q = 0;
    for (int rb = wave; rb < b / 16; rb += 16, ++q) {
      const int i0 = rb * 16;
      double bf[NLOC / 4];
#pragma unroll
      for (int kk = 0; kk < NLOC / 4; ++kk)
        bf[kk] = pre ? pre[q < VRB ? q : 0][kk] : P[(size_t)(kk * 4 + (lane >> 4)) * bp + i0 + (lane & 15)];
      jd4 acc[NT16];
#pragma unroll
      for (int tc = 0; tc < NT16; ++tc) {
        acc[tc] = jd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NLOC / 4; ++kk) {
          const double a = Wm[(kk * 4 + (lane >> 4)) * NLOC + tc * 16 + (lane & 15)];
          acc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[kk], acc[tc], 0, 0, 0);
        }
      }
#pragma unroll
      for (int tc = 0; tc < NT16; ++tc)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int gc = gcol(tc * 16 + (lane >> 4) + 4 * reg);
          if (gc < b) M[(size_t)gc * ldb + i0 + (lane & 15)] = acc[tc][reg];
        }
    }
  };

  load_panel(B);
  // The V panel is wanted only after the small eigenproblem is solved; its loads are issued now (registers) so that
  // their L2 / fabric latency -- the panels were last written by other XCDs -- passes behind the Gram product and
  // the Jacobi rounds instead of in front of the second apply.  b <= 512 (two row blocks per wave), else loaded late as before.
  // They are fetched in the layout the second apply multiplies them in (the wave's own 16-row blocks as MFMA operands: lane
  // (fr, fk) holds row i0 + fr of local columns 4 kk + fk), so the V panel never sees LDS: no store, no barrier, no reads.
  const bool v_early = b <= 16 * 16 * VRB;
  double vfr[VRB][NLOC / 4];
  if (v_early) {
#pragma unroll
    for (int q = 0; q < VRB; ++q) {
      const int i0 = (wave + 16 * q) * 16;
#pragma unroll
      for (int kk = 0; kk < NLOC / 4; ++kk) {
        const int gc = gcol(kk * 4 + (lane >> 4));
        vfr[q][kk] = (gc < b && i0 < b) ? V[(size_t)gc * ldb + i0 + (lane & 15)] : 0.0;
      }
    }
  }
  __syncthreads();
  jac_stamp(tr, 1);
  // ---- Gram block on the matrix cores
  if (wave < TILES * KP) {
    const int tile = wave % TILES, kp = wave / TILES;
    const int tp = tile / NT16, tq = tile % NT16;
    const int rows = b / KP;  // multiple of 4 because b % 16 == 0
    jd4 acc = jd4{0.0, 0.0, 0.0, 0.0};
    const double *pa = P + (size_t)(tp * 16 + (lane & 15)) * bp + (lane >> 4);
    const double *pb = P + (size_t)(tq * 16 + (lane & 15)) * bp + (lane >> 4);
    for (int i0 = kp * rows; i0 < (kp + 1) * rows; i0 += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[i0], pb[i0], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      part[(size_t)(kp * NLOC + tp * 16 + (lane >> 4) + 4 * reg) * NLOC + tq * 16 + (lane & 15)] = acc[reg];
  }
  __syncthreads();
  for (int e = tid; e < NLOC * NLOC; e += 1024) {
    double g = part[e];
#pragma unroll
    for (int kp = 1; kp < KP; ++kp) g += part[kp * NLOC * NLOC + e];
    Gm[e] = g;
    Wm[e] = (e / NLOC == e % NLOC) ? 1.0 : 0.0;
  }
  __syncthreads();
  jac_stamp(tr, 2);
  // ---- two-sided cyclic Jacobi on Gm, rotations accumulated in Wm.
  // Per round: NP leader lanes turn (Gm_pp, Gm_qq, Gm_pq) into rotation coefficients, then the threads rebuild
  // J^T Gm J and Wm J from the old buffers into the other buffers (no read-modify-write), by 2 x 2 blocks (below);
  // a round costs two barriers.
  double *G2 = part;                 // the Gram partials are dead by now: reuse as the second buffers
  double *W2 = part + NLOC * NLOC;
  double *cc = part + 2 * NLOC * NLOC;   // per index: cos, signed sin; per pair of the round (= per leader): p | q << 16
  double *dd = cc + NLOC;                // (the pairs' (cos, sin) packed as one 16-byte entry per pair, read with
  int *pr = (int *)(dd + NLOC);          //  ds_read_b128: 16.5 us per visit against 11.5 -- measured, not understood)
  double *Gc = Gm, *Gn = G2, *Wc = Wm, *Wn = W2;
  const double tol2 = tol * tol;
  int rotations = 0;
  double mx_num = 0.0, mx_den = 1.0;   // largest squared cosine between two columns rotated in this visit, as a fraction:
                                       // compared by cross-multiplication, divided once at the end (an fp64 division is ~25
                                       // dependent instructions on the round's critical chain; the value only feeds a log line)
  for (int ls = 0; ls < local_sweeps; ++ls) {
    if (tid == 0) any_rot = 0;
    __syncthreads();
    int rot_here = 0;
    // cross_only: only the WB x WB pairs (p in block column I, q in block column J) are visited, in WB
    // rounds; the pairs inside a block column are left to the one visit per sweep that runs the full
    // round-robin (jacobi_run: round 0), so a sweep rotates every pair of the b columns exactly once
    const int nrounds = cross_only ? NP : NLOC - 1;
    for (int rr = 0; rr < nrounds; ++rr) {
      if (tid < NP) {
        int p, q;
        if (cross_only) { p = tid; q = NP + ((tid + rr) & (NP - 1)); }
        else rr_pair(NLOC, rr, tid, p, q);
        const double al = Gc[p * NLOC + p], be = Gc[q * NLOC + q], ga = Gc[p * NLOC + q];
        double cs = 1.0, sn = 0.0;
        if (ga * ga > tol2 * (al * be) && al > 0.0 && be > 0.0) {
          const double dl = be - al;
          const double r1 = rsqrt_nr(dl * dl + 4.0 * (ga * ga));
          const double x2 = 0.5 + 0.5 * (__builtin_fabs(dl) * r1);
          const double r2 = rsqrt_nr(x2);
          cs = x2 * r2;
          sn = (dl >= 0.0 ? ga : -ga) * (r1 * r2);
          ++rot_here;
          round_rot[rr % 3] = 1;
          if ((ga * ga) * mx_den > mx_num * (al * be)) { mx_num = ga * ga; mx_den = al * be; }
        }
        cc[p] = cs; dd[p] = -sn;              // new_p = c old_p - s old_q
        cc[q] = cs; dd[q] = sn;               // new_q = s old_p + c old_q
        pr[tid] = p | (q << 16);              // the round's pairs, by leader
      }
      if (tid == 0) round_rot[(rr + 2) % 3] = 0;   // last read two barriers ago, next set two barriers on
      __syncthreads();
      if (!round_rot[rr % 3]) continue;            // nobody rotates: both buffers stay as they are
      // A rotation round acts on 2 x 2 blocks: the block (pair a, pair b) of the matrix and the entries (row i, pair b) of
      // Wm are rebuilt by ONE thread each -- one LDS read per output instead of four (two for Wm): with sixteen waves
      // queueing at the LDS pipe the rebuild is a matter of its LDS instructions, 224 per round element by element, 112
      // this way.  Every output is the same expression, in the same operation order, as the element-wise form.
      for (int item = tid; item < NP * NP + NLOC * NP; item += 1024) {
        if (item < NP * NP) {
          const int a = item / NP, bq = item % NP;
          const int pqa = pr[a], pqb = pr[bq];
          const int pa = pqa & 0xffff, qa = pqa >> 16, pb = pqb & 0xffff, qb = pqb >> 16;
          const double ca = cc[pa], sa = dd[qa], cb = cc[pb], sb = dd[qb];
          const double g00 = Gc[pa * NLOC + pb], g01 = Gc[pa * NLOC + qb], g10 = Gc[qa * NLOC + pb], g11 = Gc[qa * NLOC + qb];
          Gn[pa * NLOC + pb] = ca * (cb * g00 + (-sb) * g01) + (-sa) * (cb * g10 + (-sb) * g11);
          Gn[pa * NLOC + qb] = ca * (cb * g01 + sb * g00) + (-sa) * (cb * g11 + sb * g10);
          Gn[qa * NLOC + pb] = ca * (cb * g10 + (-sb) * g11) + sa * (cb * g00 + (-sb) * g01);
          Gn[qa * NLOC + qb] = ca * (cb * g11 + sb * g10) + sa * (cb * g01 + sb * g00);
        } else {
          const int v = item - NP * NP;
          const int i = v / NP, bq = v % NP;
          const int pqb = pr[bq];
          const int pb = pqb & 0xffff, qb = pqb >> 16;
          const double cb = cc[pb], sb = dd[qb];
          const double w0 = Wc[i * NLOC + pb], w1 = Wc[i * NLOC + qb];
          Wn[i * NLOC + pb] = cb * w0 + (-sb) * w1;
          Wn[i * NLOC + qb] = cb * w1 + sb * w0;
        }
      }
      __syncthreads();
      double *t1 = Gc; Gc = Gn; Gn = t1;
      double *t2 = Wc; Wc = Wn; Wn = t2;
    }
    rotations += rot_here;
    if (rot_here) { any_rot = 1; visit_rot = 1; }
    __syncthreads();
    if (!any_rot) break;
    __syncthreads();
  }
  __syncthreads();
  jac_stamp(tr, 3);
  if (tr && tid == 0) { tr[6] = cross_only; tr[7] = round; tr[4] = tr[5] = 0; trace[0] = trace[0] + 1; }
  if (!visit_rot) return;   // the NLOC columns were orthogonal to the threshold already: panels untouched
  if (Wc != Wm) {   // an odd number of rounds ran: move the result where apply_w reads it
    for (int e = tid; e < NLOC * NLOC; e += 1024) Wm[e] = Wc[e];
  }
  __syncthreads();
  if (rotations) {
    atomicAdd(&flags[0], rotations);
    atomicMax(&flags[3], __float_as_int((float)__builtin_sqrt(mx_num / mx_den) * 1.0000002f));   // >= 0: the bit patterns order like the values
  }
  // ---- apply the accumulated rotation to the B panel, then to the V panel (same LDS buffer)
  apply_w(B, nullptr);
  jac_stamp(tr, 4);
  if (v_early) {
    apply_w(V, vfr);
  } else {
    __syncthreads();        // every wave has read its rows of the B panel: the buffer may take the V panel
    load_panel(V);
    __syncthreads();
    apply_w(V, nullptr);
  }
  jac_stamp(tr, 5);
}

// ------------------------------------------------------------------------------------------
// The same visit for panels that do NOT fit LDS (b above ~1000: the full decomposition of a large Gram matrix, the
// K == s branch of truncated_SVD_cpp, src/TruncatedSVD.cpp:17-20, for s up to 16384).  The 32 columns of the pair are
// streamed through LDS in chunks of JS_CHUNK rows: pass 1 accumulates the 32 x 32 Gram block on the matrix cores (one
// accumulator per wave, carried across the chunks; the sixteen partial tiles are added in a fixed order), the small
// eigenproblem is solved in LDS as above (one full cyclic sweep of two-sided rotations), pass 2 streams the B panel and
// then the V panel again and applies the accumulated rotation by MFMA.  Per visit the B panel is read twice and
// written once, the V panel read and written once: at s = 5000 a round of 156 visits moves ~1 GB (0.2 ms), a sweep
// of 312 rounds takes ~60 ms and the decomposition converges in 8-10 sweeps -- well under a second for what the
// reference does with a dense BDCSVD of the n x s matrix.
// ------------------------------------------------------------------------------------------
constexpr int JS_CHUNK = 256;      // rows per chunk: 32 x 258 doubles of LDS

__global__ __launch_bounds__(1024) void jac_stream_kernel(double *__restrict__ B, double *__restrict__ V, int b, int ldb,
                                                          int nbc, int round, double tol, int *__restrict__ flags) {
  constexpr int NLOC = 32, WB = 16, NP = 16, NT16 = 2, TILES = 4, KP = 4;
  constexpr int CP = JS_CHUNK + 2;          // LDS column stride
  __shared__ double P[NLOC * CP];
  __shared__ double Gm[NLOC * NLOC], Wm[NLOC * NLOC], G2[NLOC * NLOC], W2[NLOC * NLOC];
  __shared__ double part[TILES * KP][16 * 16];
  __shared__ double cc[NLOC], dd[NLOC];
  __shared__ int pr[NLOC];
  __shared__ int round_rot[3];
  __shared__ int visit_rot;
  if (flags[1]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 3) round_rot[tid] = 0;
  if (tid == 0) visit_rot = 0;
  int I, J;
  rr_pair(nbc, round, blockIdx.x, I, J);
  const int cI = I * WB, cJ = J * WB;
  if (cI >= b) return;
  auto gcol = [&](int c) { return (c < WB) ? cI + c : cJ + (c - WB); };
  const int nchunk = (b + JS_CHUNK - 1) / JS_CHUNK;
  auto load_chunk = [&](const double *M, int r0) {       // rows r0 .. r0 + JS_CHUNK of the 32 columns (zeros outside the matrix)
    for (int c = wave; c < NLOC; c += 16) {
      const int gc = gcol(c);
      for (int i = lane; i < JS_CHUNK; i += 64) P[c * CP + i] = (gc < b && r0 + i < b) ? M[(size_t)gc * ldb + r0 + i] : 0.0;
    }
  };
  auto store_chunk = [&](double *M, int r0) {
    for (int c = wave; c < NLOC; c += 16) {
      const int gc = gcol(c);
      if (gc < b)
        for (int i = lane; i < JS_CHUNK; i += 64)
          if (r0 + i < b) M[(size_t)gc * ldb + r0 + i] = P[c * CP + i];
    }
  };
  // ---- pass 1: Gram block; wave = (tile, row part), each part a quarter of every chunk
  {
    const int tile = wave % TILES, kp = wave / TILES;
    const int tp = tile / NT16, tq = tile % NT16;
    jd4 acc = jd4{0.0, 0.0, 0.0, 0.0};
    const double *pa = P + (tp * 16 + (lane & 15)) * CP + (lane >> 4);
    const double *pb = P + (tq * 16 + (lane & 15)) * CP + (lane >> 4);
    constexpr int rows = JS_CHUNK / KP;
    for (int ch = 0; ch < nchunk; ++ch) {
      __syncthreads();
      load_chunk(B, ch * JS_CHUNK);
      __syncthreads();
      for (int i0 = kp * rows; i0 < (kp + 1) * rows; i0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[i0], pb[i0], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[wave][((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[reg];
  }
  __syncthreads();
  for (int e = tid; e < NLOC * NLOC; e += 1024) {
    const int i = e / NLOC, j = e % NLOC;
    const int tile = (i / 16) * NT16 + (j / 16), w = (i % 16) * 16 + (j % 16);
    double g = part[tile][w];
#pragma unroll
    for (int kp = 1; kp < KP; ++kp) g += part[kp * TILES + tile][w];
    Gm[e] = g;
    Wm[e] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- one cyclic sweep of two-sided rotations on the 32 x 32 block (as in jac_block_kernel)
  double *Gc = Gm, *Gn = G2, *Wc = Wm, *Wn = W2;
  const double tol2 = tol * tol;
  int rotations = 0;
  for (int rr = 0; rr < NLOC - 1; ++rr) {
    if (tid < NP) {
      int p, q;
      rr_pair(NLOC, rr, tid, p, q);
      const double al = Gc[p * NLOC + p], be = Gc[q * NLOC + q], ga = Gc[p * NLOC + q];
      double cs = 1.0, sn = 0.0;
      if (ga * ga > tol2 * (al * be) && al > 0.0 && be > 0.0) {
        const double dl = be - al;
        const double r1 = rsqrt_nr(dl * dl + 4.0 * (ga * ga));
        const double x2 = 0.5 + 0.5 * (__builtin_fabs(dl) * r1);
        const double r2 = rsqrt_nr(x2);
        cs = x2 * r2;
        sn = (dl >= 0.0 ? ga : -ga) * (r1 * r2);
        ++rotations;
        round_rot[rr % 3] = 1;
        visit_rot = 1;
      }
      cc[p] = cs; dd[p] = -sn; pr[p] = q;
      cc[q] = cs; dd[q] = sn;  pr[q] = p;
    }
    if (tid == 0) round_rot[(rr + 2) % 3] = 0;
    __syncthreads();
    if (!round_rot[rr % 3]) continue;
    for (int e = tid; e < NLOC * NLOC; e += 1024) {
      const int i = e / NLOC, j = e % NLOC;
      const int pi = pr[i], pj = pr[j];
      const double ci = cc[i], di = dd[i], cj = cc[j], dj = dd[j];
      Gn[e] = ci * (cj * Gc[i * NLOC + j] + dj * Gc[i * NLOC + pj]) + di * (cj * Gc[pi * NLOC + j] + dj * Gc[pi * NLOC + pj]);
      Wn[e] = cj * Wc[i * NLOC + j] + dj * Wc[i * NLOC + pj];
    }
    __syncthreads();
    double *t1 = Gc; Gc = Gn; Gn = t1;
    double *t2 = Wc; Wc = Wn; Wn = t2;
  }
  __syncthreads();
  if (!visit_rot) return;          // the 32 columns were orthogonal to the threshold already
  if (Wc != Wm)
    for (int e = tid; e < NLOC * NLOC; e += 1024) Wm[e] = Wc[e];
  __syncthreads();
  if (tid < NP && rotations) atomicAdd(&flags[0], rotations);
  // ---- pass 2: P <- P Wm, chunk by chunk, for the B panel and then the V panel
  auto apply_w = [&]() {
    for (int rb = wave; rb < JS_CHUNK / 16; rb += 16) {
      const int i0 = rb * 16;
      double bf[NLOC / 4];
#pragma unroll
      for (int kk = 0; kk < NLOC / 4; ++kk) bf[kk] = P[(kk * 4 + (lane >> 4)) * CP + i0 + (lane & 15)];
      jd4 acc2[NT16];
#pragma unroll
      for (int tc = 0; tc < NT16; ++tc) {
        acc2[tc] = jd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NLOC / 4; ++kk) {
          const double a = Wm[(kk * 4 + (lane >> 4)) * NLOC + tc * 16 + (lane & 15)];
          acc2[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[kk], acc2[tc], 0, 0, 0);
        }
      }
#pragma unroll
      for (int tc = 0; tc < NT16; ++tc)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) P[(tc * 16 + (lane >> 4) + 4 * reg) * CP + i0 + (lane & 15)] = acc2[tc][reg];
    }
  };
  for (int which = 0; which < 2; ++which) {
    double *M = which == 0 ? B : V;
    for (int ch = 0; ch < nchunk; ++ch) {
      __syncthreads();
      load_chunk(M, ch * JS_CHUNK);
      __syncthreads();
      apply_w();
      __syncthreads();
      store_chunk(M, ch * JS_CHUNK);
    }
  }
}

// ------------------------------------------------------------------------------------------
// b x b x b products of the orthonormalisation (Newton-Schulz: three per iteration, hundreds per
// solve).  The tiled GEMM of gemm.hip needs split-K and a reduction launch to find 33 MFLOP of
// parallelism in four 128 x 128 tiles (14 + 9 us); here every 16 x 16 tile of C is one wave, operands go
// straight from L2 into the MFMA operand layout (no LDS, no reduction):
//   C = alpha A B + beta E   (all b x b, column-major, leading dimension b; b % 16 == 0)
// Lane l = (r = l & 15, kq = l >> 4) owns the contiguous k range [kq b/4, (kq+1) b/4): step t feeds
// A(i0 + r, kq b/4 + t) and B(kq b/4 + t, j0 + r) -- a permutation of the k index, the same for both
// operands.  Up to two independent products per launch (blockIdx.z).
// ------------------------------------------------------------------------------------------
struct SmallGemm {
  const double *A, *B, *E;
  double *C;
  double alpha, beta;
};
struct SmallGemmPair {
  SmallGemm g[2];
};

// (one wave per SIMD: the whole register file is this wave's)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void small_gemm_kernel(SmallGemmPair args, int b) {
  const SmallGemm g = args.g[blockIdx.z];
  const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  const int kn = b >> 2;
  const double *pa = g.A + (size_t)(kq * kn) * b + i0 + r;      // A(i0 + r, kq kn + t): stride b in t
  const double *pb = g.B + (size_t)(j0 + r) * b + kq * kn;      // B(kq kn + t, j0 + r): contiguous in t
  // (Four accumulators in turn instead of one chain of b / 4 dependent MFMAs, round 3: 7.20 against 7.03 us per launch --
  //  the two round trips for the operands and the store are the launch, not the 1.9 us of MFMAs.)
  jd4 acc = jd4{0.0, 0.0, 0.0, 0.0};
  int t = 0;
  // the launch has one wave per SIMD at most (b / 16 squared waves), so registers are free: 32 steps of operands in
  // flight at a time -- with 8 the kernel was eight round trips to L2 long (7.5 us for 2 us of MFMAs)
  // B's operands two k at a time (16 bytes per lane): a lane walks its own column of B, so every load instruction touches 64
  // different cache lines whatever its width -- half as many of them (measured: 7.07 us per launch against 7.0, i.e. this
  // is not what the launch waits for either)
  // (b = 256: all 64 steps of operands in flight at once -- one round trip to L2 instead of two; the wave is alone on its SIMD
  //  and may use the whole register file)
  for (; t + 64 <= kn; t += 64) {
    double av[64];
    d2v bv2[32];
#pragma unroll
    for (int u = 0; u < 64; ++u) av[u] = pa[(size_t)(t + u) * b];
#pragma unroll
    for (int u = 0; u < 32; ++u) bv2[u] = *(const d2v *)(pb + t + 2 * u);
    __builtin_amdgcn_sched_barrier(0);   // (left to itself the scheduler sinks the loads between the MFMAs: a dozen in flight, 52 VGPRs)
#pragma unroll
    for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv2[u >> 1][u & 1], acc, 0, 0, 0);
  }
  for (; t + 32 <= kn; t += 32) {
    double av[32];
    d2v bv2[16];
#pragma unroll
    for (int u = 0; u < 32; ++u) av[u] = pa[(size_t)(t + u) * b];
#pragma unroll
    for (int u = 0; u < 16; ++u) bv2[u] = *(const d2v *)(pb + t + 2 * u);
#pragma unroll
    for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv2[u >> 1][u & 1], acc, 0, 0, 0);
  }
  for (; t + 8 <= kn; t += 8) {
    double av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { av[u] = pa[(size_t)(t + u) * b]; bv[u] = pb[t + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  for (; t < kn; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)t * b], pb[t], acc, 0, 0, 0);
  // D(row = kq + 4 reg, col = r)
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const size_t e = (size_t)(j0 + r) * b + i0 + kq + 4 * reg;
    double v = g.alpha * acc[reg];
    if (g.E) v += g.beta * g.E[e];
    g.C[e] = v;
  }
}

// Symmetric eigendecomposition of a small matrix (g <= N) by one workgroup: classical two-sided
// cyclic Jacobi, the whole matrix and the rotation product in LDS, every round = NP leader lanes
// computing rotations + every thread rebuilding its elements of J^T A J and W J into the other
// buffers (two barriers per round).  Used for the dense guard block of the Rayleigh-Ritz matrix.
template <int N>
__global__ __launch_bounds__(1024) void small_sym_eig_kernel(const double *__restrict__ T, int ldt, int g,
                                                             double *__restrict__ Vout, double tol, int max_sweeps) {
  constexpr int NP = N / 2;
  extern __shared__ double sm[];
  __shared__ int any_rot;
  __shared__ int round_rot[3];
  double *Gc = sm, *Gn = sm + N * N, *Wc = sm + 2 * N * N, *Wn = sm + 3 * N * N;
  double *cc = sm + 4 * N * N, *dd = cc + N;
  int *pr = (int *)(dd + N);
  const int tid = threadIdx.x;
  if (tid < 3) round_rot[tid] = 0;
  for (int e = tid; e < N * N; e += 1024) {
    const int i = e / N, j = e % N;
    Gc[e] = (i < g && j < g) ? 0.5 * (T[(size_t)j * ldt + i] + T[(size_t)i * ldt + j]) : 0.0;
    Wc[e] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double tol2 = tol * tol;
  for (int sw = 0; sw < max_sweeps; ++sw) {
    if (tid == 0) any_rot = 0;
    __syncthreads();
    int rot_here = 0;
    for (int rr = 0; rr < N - 1; ++rr) {
      if (tid < NP) {
        int p, q;
        rr_pair(N, rr, tid, p, q);
        const double al = Gc[p * N + p], be = Gc[q * N + q], ga = Gc[p * N + q];
        double cs = 1.0, sn = 0.0;
        if (ga * ga > tol2 * __builtin_fabs(al * be) && ga != 0.0) {
          const double dl = be - al;
          const double r1 = rsqrt_nr(dl * dl + 4.0 * (ga * ga));
          const double x2 = 0.5 + 0.5 * (__builtin_fabs(dl) * r1);
          const double r2 = rsqrt_nr(x2);
          cs = x2 * r2;
          sn = (dl >= 0.0 ? ga : -ga) * (r1 * r2);
          ++rot_here;
          round_rot[rr % 3] = 1;
        }
        cc[p] = cs; dd[p] = -sn;
        cc[q] = cs; dd[q] = sn;
        pr[tid] = p | (q << 16);              // the round's pairs, by leader
      }
      if (tid == 0) round_rot[(rr + 2) % 3] = 0;
      __syncthreads();
      if (!round_rot[rr % 3]) continue;
      // rebuild by 2 x 2 blocks (see jac_block_kernel): the same expressions, half the LDS instructions
      for (int item = tid; item < NP * NP + N * NP; item += 1024) {
        if (item < NP * NP) {
          const int a = item / NP, bq = item % NP;
          const int pqa = pr[a], pqb = pr[bq];
          const int pa = pqa & 0xffff, qa = pqa >> 16, pb = pqb & 0xffff, qb = pqb >> 16;
          const double ca = cc[pa], sa = dd[qa], cb = cc[pb], sb = dd[qb];
          const double g00 = Gc[pa * N + pb], g01 = Gc[pa * N + qb], g10 = Gc[qa * N + pb], g11 = Gc[qa * N + qb];
          Gn[pa * N + pb] = ca * (cb * g00 + (-sb) * g01) + (-sa) * (cb * g10 + (-sb) * g11);
          Gn[pa * N + qb] = ca * (cb * g01 + sb * g00) + (-sa) * (cb * g11 + sb * g10);
          Gn[qa * N + pb] = ca * (cb * g10 + (-sb) * g11) + sa * (cb * g00 + (-sb) * g01);
          Gn[qa * N + qb] = ca * (cb * g11 + sb * g10) + sa * (cb * g01 + sb * g00);
        } else {
          const int v = item - NP * NP;
          const int i = v / NP, bq = v % NP;
          const int pqb = pr[bq];
          const int pb = pqb & 0xffff, qb = pqb >> 16;
          const double cb = cc[pb], sb = dd[qb];
          const double w0 = Wc[i * N + pb], w1 = Wc[i * N + qb];
          Wn[i * N + pb] = cb * w0 + (-sb) * w1;
          Wn[i * N + qb] = cb * w1 + sb * w0;
        }
      }
      __syncthreads();
      double *t1 = Gc; Gc = Gn; Gn = t1;
      double *t2 = Wc; Wc = Wn; Wn = t2;
    }
    if (rot_here) any_rot = 1;
    __syncthreads();
    if (!any_rot) break;
    __syncthreads();
  }
  for (int e = tid; e < g * g; e += 1024) {
    const int i = e % g, j = e / g;
    Vout[e] = Wc[i * N + j];   // column-major g x g: column j = eigenvector j
  }
}

// keep the strictly upper triangle (i < j) of the b x b matrix C, zero the rest
__global__ void mask_strict_upper_kernel(double *__restrict__ C, int b, int lower = 0) {   // lower: keep i > j instead
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * b) return;
  const int i = (int)(e % b), j = (int)(e / b);
  if (lower ? i <= j : i >= j) C[e] = 0.0;
}

// V0 = blockdiag(I_K, Vg): identity with the g x g block Vg in its lower-right corner
__global__ void embed_block_kernel(const double *__restrict__ Vg, int g, int K, int b, double *__restrict__ V0) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * b) return;
  const int i = (int)(e % b), j = (int)(e / b);
  double v = (i == j) ? 1.0 : 0.0;
  if (i >= K && j >= K) v = Vg[(size_t)(j - K) * g + (i - K)];
  V0[e] = v;
}

// after a sweep: latch convergence (no rotation applied) and reset the counter
__global__ void jac_sweep_end_kernel(int *__restrict__ flags) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (flags[0] == 0) flags[1] = 1;
    if (flags[2] < 8) flags[4 + flags[2]] = flags[3];   // largest cosine rotated in the sweep (a float's bits), kept for the log
    flags[3] = 0;
    flags[2] += 1;       // sweeps run (only counts while not converged)
    flags[0] = 0;
  }
}

// eigenvalue j = v_j . (T v_j) = v_j . B_j  (B = T V)
__global__ void jac_values_kernel(const double *__restrict__ B, const double *__restrict__ V, int b, int ldb,
                                  double *__restrict__ lam) {
  const int j = blockIdx.x;
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < b; i += blockDim.x) acc += V[(size_t)j * ldb + i] * B[(size_t)j * ldb + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) lam[j] = red[0];
}

// W(i, j) = rowscale[i] * V(i, perm[j]) * scale[j]   (b x ncols)
__global__ void permute_scale_kernel(const double *__restrict__ V, int ldv, int b, const int *__restrict__ perm,
                                     const double *__restrict__ scale, const double *__restrict__ rowscale,
                                     int ncols, double *__restrict__ W, int ldw, int transposed = 0) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * ncols) return;
  const int i = (int)(e % b), j = (int)(e / b);
  double v = V[(size_t)perm[j] * ldv + i];
  if (scale) v *= scale[j];
  if (rowscale) v *= rowscale[i];
  W[transposed ? (size_t)i * ldw + j : (size_t)j * ldw + i] = v;      // transposed: W k-major, for rot.hip
}

// Ritz values in descending order without the host: perm[rank] = j, sorted[rank] = lam[j], rank = number of values
// ahead of lam[j] (larger, or equal with a lower index: the order std::stable_sort gives)
__global__ void ritz_sort_kernel(const double *__restrict__ lam, int b, int *__restrict__ perm, double *__restrict__ sorted) {
  // (the values pass through LDS 256 at a time: as a loop over global memory every thread made b dependent trips to L2, 15 us at b = 256)
  __shared__ double sl[256];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const double mine = j < b ? lam[j] : 0.0;
  int rank = 0;
  for (int i0 = 0; i0 < b; i0 += 256) {
    __syncthreads();
    if ((int)threadIdx.x < 256 && i0 + (int)threadIdx.x < b) sl[threadIdx.x] = lam[i0 + threadIdx.x];
    __syncthreads();
    const int cnt = (b - i0 < 256) ? b - i0 : 256;
    for (int q = 0; q < cnt; ++q) {
      const double v = sl[q];
      rank += (v > mine) || (v == mine && i0 + q < j);
    }
  }
  if (j >= b) return;
  perm[rank] = j;
  sorted[rank] = mine;
}

// S <- D S D with D = diag(1/sqrt(S_jj)): the Gram matrix of the column-normalised block
__global__ void sym_scale_diag_kernel(const double *__restrict__ S, int b, double *__restrict__ dinv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b) return;
  const double d = S[(size_t)j * b + j];
  dinv[j] = d > 0.0 ? 1.0 / __builtin_sqrt(d) : 0.0;
}
__global__ void sym_scale_apply_kernel(double *__restrict__ S, int b, const double *__restrict__ dinv) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * b) return;
  const int i = (int)(e % b), j = (int)(e / b);
  S[e] = S[e] * dinv[i] * dinv[j];
}

// out[blk] = partial sums of |S - I|_F^2: DIST_BLOCKS blocks, fixed assignment and order (the host adds them)
constexpr int DIST_BLOCKS = 32;
__global__ __launch_bounds__(256) void dist_to_identity_kernel(const double *__restrict__ S, int b, double *__restrict__ out,
                                                               double diag = 1.0) {   // diag = 0: |S|_F^2
  __shared__ double red[256];
  double acc = 0.0;
  const long tot = (long)b * b;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)DIST_BLOCKS * 256) {
    const int i = (int)(e % b), j = (int)(e / b);
    const double d = S[e] - (i == j ? diag : 0.0);
    acc = __builtin_fma(d, d, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x] = red[0]; __threadfence_system(); }   // (out may be host memory)
}

__global__ void set_identity_kernel(double *__restrict__ M, int b) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * b) return;
  M[e] = ((int)(e % b) == (int)(e / b)) ? 1.0 : 0.0;
}

// W(i,j) = rowscale[i] * Z(i,j)
__global__ void row_scale_kernel(const double *__restrict__ Z, int b, const double *__restrict__ rowscale,
                                 double *__restrict__ W, double pre = 1.0, int transposed = 0) {   // pre: a scalar factor applied first
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)b * b) return;
  const int i = (int)(e % b), j = (int)(e / b);
  W[transposed ? (size_t)i * b + j : (size_t)e] = (pre * Z[e]) * rowscale[i];      // transposed: W k-major, for rot.hip
}

// res[j] = | Z(:,j) - theta_j Q(:,j) |_2 , j < K ; also column norms of Q when Z == nullptr
// (host: optional host-visible copy of what the step's one round trip asks for -- [0, gridDim.x) the residuals, [b, 2b) theta)
__global__ void resid_kernel(const double *__restrict__ Z, const double *__restrict__ Q, int s, int ld,
                             const double *__restrict__ theta, double *__restrict__ res, double *__restrict__ host, int b,
                             int tstride) {   // tstride = b + 1: theta_j = the diagonal of a b x b matrix (and no copy of theta to the host)
  const int j = blockIdx.x;
  __shared__ double red[256];
  double acc = 0.0;
  const double th = Z ? theta[(size_t)j * tstride] : 0.0;
#pragma unroll 8     // (eight iterations' loads in flight; the chain keeps its order)
  for (int i = threadIdx.x; i < s; i += blockDim.x) {
    const double q = Q[(size_t)j * ld + i];
    const double d = Z ? Z[(size_t)j * ld + i] - th * q : q;
    acc = __builtin_fma(d, d, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) res[j] = __builtin_sqrt(red[0]);
  if (host) {
    if (threadIdx.x == 0) host[j] = __builtin_sqrt(red[0]);
    if (tstride == 1)
      for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < b; q += gridDim.x * blockDim.x) host[b + q] = theta[q];
    __threadfence_system();
  }
}

// A-priori spectrum bounds of the symmetric PSD matrix G, for the one filter that runs before any Rayleigh-Ritz step:
// out[blk] = max over the columns of block blk of the absolute column sum (>= lambda_max: the 1-norm), out[nblk + blk] =
// the block's share of the trace.  One wave per column, APRIORI_BLOCKS workgroups of four waves.
constexpr int APRIORI_BLOCKS = 256;
__global__ __launch_bounds__(256) void apriori_bounds_kernel(const double *__restrict__ G, int ldg, int s,
                                                             double *__restrict__ out) {
  __shared__ double wmax[4], wtr[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double cmax = 0.0, tr = 0.0;
  for (int j = blockIdx.x * 4 + wave; j < s; j += APRIORI_BLOCKS * 4) {
    const double *col = G + (size_t)j * ldg;
    double a = 0.0;
    for (int i = lane; i < s; i += 64) a += __builtin_fabs(col[i]);
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    cmax = a > cmax ? a : cmax;
    tr += col[j];
  }
  if (lane == 0) { wmax[wave] = cmax; wtr[wave] = tr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = wmax[0], t = wtr[0];
    for (int q = 1; q < 4; ++q) { m = wmax[q] > m ? wmax[q] : m; t += wtr[q]; }
    out[blockIdx.x] = m;
    out[APRIORI_BLOCKS + blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
struct JacobiPlan { int w, nbc, nt; size_t lds; int nloc; /* 32 / 16: MFMA block kernel, 0: scalar kernel */ };

static size_t jac_block_lds(int b, int nloc) { return sizeof(double) * ((size_t)nloc * (b + 2) + 6 * (size_t)nloc * nloc); }
static_assert(true, "part[] holds 4 Gram partials = 4 nloc^2 doubles; the local Jacobi reuses 2 nloc^2 + 3 nloc of them");

static JacobiPlan jacobi_plan(int b) {
  JacobiPlan p;
  p.nloc = 0;
  if (b % 16 == 0 && b >= 32 && tuning("jacobi_scalar", 0) == 0) {
    for (int nloc : {32, 16}) {
      if (nloc == 32 && tuning("jacobi_nloc", 32) == 16 && b >= 64) continue;      // (experiments: smaller column blocks, more of them)
      if (jac_block_lds(b, nloc) <= 150 * 1024) {
        p.nloc = nloc; p.w = nloc / 2; p.nbc = (b + p.w - 1) / p.w;
        if (p.nbc & 1) ++p.nbc;
        p.nt = 1024; p.lds = jac_block_lds(b, nloc);
        return p;
      }
    }
  }
  if (b > tuning("jacobi_stream_above", 1024) && tuning("jacobi_scalar", 0) == 0) {
    // panels too long for LDS: streamed visits (jac_stream_kernel); any b (ragged rows and block columns are handled)
    p.nloc = -1; p.w = 16; p.nbc = (b + 15) / 16;
    if (p.nbc & 1) ++p.nbc;
    p.nt = 1024; p.lds = 0;
    return p;
  }
  // 2w columns of B and of V, each b+16 doubles, must fit ~150 KB of LDS
  int w = (int)((150 * 1024) / (sizeof(double) * 4 * (size_t)(b + 16)));
  int pw = 1;
  while (pw * 2 <= w && pw < 32) pw *= 2;
  p.w = pw;
  p.nbc = (b + p.w - 1) / p.w;
  if (p.nbc & 1) ++p.nbc;
  if (p.nbc < 2) p.nbc = 2;
  p.nt = 1024;
  p.lds = sizeof(double) * 4 * (size_t)p.w * (b + 16);
  return p;
}

struct EigWork {
  // big (s x b) buffers
  double *Q, *Y, *Yp, *Z, *Qold;
  // small (b x b)
  double *T, *JB, *JV, *W, *X2, *Id;
  double *lam, *scale, *res, *theta, *dinv, *gemm_ws, *apriori;   // res (b) and theta (b) are adjacent: one copy to the host
  int *perm, *flags;
  int *tickets;         // GEMM_MAX_TICKETS zeroed counters of the in-kernel split-K reduction (main stream only)
  double *red;          // one double per 16 x 16 tile of a b x b matrix + an int counter (zero between launches): GemmFusedReduce
  int *redcnt;
  size_t gemm_ws_elems;
};

static size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

static int eig_block_size(int s, int K) {
  int guard = K * tuning("eig_guard_pct", 25) / 100;
  if (guard < 24) guard = 24;
  int b = K + guard;
  b = (b + 15) / 16 * 16;
  if (b > s) b = s;
  return b;
}

static bool eig_use_dense(int s, int K) {
  const int b = eig_block_size(s, K);
  return K >= s || s <= 256 || 2 * b >= s;
}

static size_t eig_gemm_ws_elems(int s, int b) {
  // split-K partials (whole 128 x 128 tiles): the s x b products use up to 16 planes, the b x b ones up to 128
  const size_t sp = (size_t)(s + 127) / 128 * 128, bp = (size_t)(b + 127) / 128 * 128;
  size_t a = (size_t)16 * sp * bp, c = (size_t)128 * bp * bp;
  return a > c ? a : c;
}

// Block-sparse products with G: bsg.h / bsg.hip.  The two kernels below move blocks in and out of its transposed layout.

// tiled transpose: in is R x C with element (r, c) at r + c*R; out(c, r) at c + r*C
__global__ void bs_transpose_kernel(const double *__restrict__ in, int R, int C, double *__restrict__ out) {
  __shared__ double t[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
  for (int cc = ty; cc < 32; cc += 8) {
    const int r = r0 + tx, c = c0 + cc;
    t[cc][tx] = (r < R && c < C) ? in[(size_t)c * R + r] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    if (r < R && c < C) out[(size_t)r * C + c] = t[tx][rr];
  }
}
// V(perm[i'], k) = R(i', k)
__global__ void bs_unpermute_kernel(const double *__restrict__ R, int s, int K, const int *__restrict__ perm,
                                    double *__restrict__ V, int ldv) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * K) return;
  const int ip = (int)(e % s), k = (int)(e / s);
  V[(size_t)k * ldv + perm[ip]] = R[e];
}

static size_t eig_workspace_bytes(int s, int K) {
  const bool dense = eig_use_dense(s, K);
  const int b = dense ? s : eig_block_size(s, K);
  size_t tot = 0;
  if (!dense) tot += 5 * align_up(sizeof(double) * (size_t)s * b);
  tot += 6 * align_up(sizeof(double) * (size_t)b * b);
  tot += 5 * align_up(sizeof(double) * (size_t)b) + align_up(sizeof(double) * 2 * APRIORI_BLOCKS);
  tot += align_up(sizeof(int) * (size_t)b) + align_up(sizeof(int) * 16) + align_up(sizeof(int) * GEMM_MAX_TICKETS);
  if (!dense) tot += align_up(sizeof(double) * eig_gemm_ws_elems(s, b));
  if (!dense) tot += align_up(sizeof(double) * ((size_t)ceil_div(b, 16) * ceil_div(b, 16) + 2));
  if (!dense && s >= 1024) tot += bsg_workspace_bytes(s, b);   // (used from s = 1536 on by default; tunable)
  return tot + 1024;
}

// symmetric eigendecomposition of the b x b matrix T (device): on return JV holds eigenvectors,
// h_lam the eigenvalues (unsorted, host copy).  Synchronises the stream.
// Runs global sweeps on the pair (JB, JV) that is already set up (JB = T JV, JV orthogonal).
// sweep_limit < 0: iterate until a whole sweep applies no rotation above `tol_scale` x the
// rounding threshold (at most 60 sweeps; `strict` turns a miss into FLGP_ERR_NOCONV);
// sweep_limit = k > 0: exactly k sweeps, a refinement step on an already nearly diagonal T.
// JV is a product of plane rotations, i.e. orthogonal to rounding, however early the loop stops.
static int jacobi_run(hipStream_t st, int b, EigWork &w, std::vector<double> &h_lam, int *sweeps_out,
                      int sweep_limit, double tol_scale, bool strict, double *JB = nullptr, double *JV = nullptr,
                      bool to_host = true) {
  if (!JB) JB = w.JB;
  if (!JV) JV = w.JV;
  const JacobiPlan p = jacobi_plan(b);
  const void *kfn = p.nloc == 32 ? (const void *)jac_block_kernel<32>
                  : p.nloc == 16 ? (const void *)jac_block_kernel<16> : (const void *)jac_round_kernel;
  if (p.nloc >= 0 && p.lds > 48 * 1024)
    FLGP_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
  FLGP_HIP(hipMemsetAsync(w.flags, 0, sizeof(int) * 12, st));
  const bool to_convergence = sweep_limit < 0;
  const int max_sweeps = to_convergence ? 60 : sweep_limit;
  // rotate while |b_p . b_q| > tol |b_p||b_q|; a dot product of length b carries ~sqrt(b) eps of
  // rounding noise, so a tighter threshold only chases noise (cf. LAPACK dgesvj: sqrt(m) eps)
  const double tol = tol_scale * 4.0 * std::sqrt((double)b) * 1.1102230246251565e-16;
  int h_flags[4] = {0, 0, 0, 0};
  const int cross_from = tuning("jacobi_cross_from", 0);   // 1000 = never
  for (int sw = 0; sw < max_sweeps; ++sw) {
    for (int round = 0; round < p.nbc - 1; ++round) {
      const int cross = (round > 0 && sw >= cross_from) ? 1 : 0;
      if (p.nloc < 0)
        hipLaunchKernelGGL(jac_stream_kernel, dim3(p.nbc / 2), dim3(1024), 0, st, JB, JV, b, b, p.nbc, round, tol, w.flags);
      else if (p.nloc == 32)
        hipLaunchKernelGGL(jac_block_kernel<32>, dim3(p.nbc / 2), dim3(1024), p.lds, st, JB, JV, b, b, p.nbc, round,
                           tol, w.flags, tuning("jacobi_local_sweeps", 1), cross, g_jac_trace, tuning("jacobi_batched_load", 1));
      else if (p.nloc == 16)
        hipLaunchKernelGGL(jac_block_kernel<16>, dim3(p.nbc / 2), dim3(1024), p.lds, st, JB, JV, b, b, p.nbc, round,
                           tol, w.flags, tuning("jacobi_local_sweeps", 1), cross, g_jac_trace, tuning("jacobi_batched_load", 1));
      else
        hipLaunchKernelGGL(jac_round_kernel, dim3(p.nbc / 2), dim3(p.nt), p.lds, st, JB, JV, b, b, p.w, p.nbc,
                           round, tol, w.flags, 1);
    }
    hipLaunchKernelGGL(jac_sweep_end_kernel, dim3(1), dim3(64), 0, st, w.flags);
    FLGP_TRY(check_launch("jac_round_kernel"));
    if (to_convergence && sw >= 2) {  // from the third sweep on, ask the device whether it is done
      FLGP_HIP(hipMemcpyAsync(h_flags, w.flags, sizeof(int) * 3, hipMemcpyDeviceToHost, st));
      FLGP_HIP(stream_wait(st));
      if (h_flags[1]) break;
    }
  }
  hipLaunchKernelGGL(jac_values_kernel, dim3(b), dim3(256), 0, st, JB, JV, b, b, w.lam);
  FLGP_TRY(check_launch("jac_values_kernel"));
  if (tuning("eig_verbose", 0) > 2) {
    int hf[12];
    FLGP_HIP(hipMemcpyAsync(hf, w.flags, sizeof(hf), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "[flgp jacobi] b=%d sweeps=%d largest column cosine rotated, per sweep:", b, hf[2]);
    for (int q = 0; q < hf[2] && q < 8; ++q) { float f; memcpy(&f, &hf[4 + q], 4); fprintf(stderr, " %.2e", f); }
    fprintf(stderr, "\n");
  }
  if (!to_host) return FLGP_OK;        // a fixed number of sweeps on the device: the eigenvalues stay in w.lam
  h_lam.resize(b);
  FLGP_HIP(hipMemcpyAsync(h_lam.data(), w.lam, sizeof(double) * b, hipMemcpyDeviceToHost, st));
  FLGP_HIP(hipMemcpyAsync(h_flags, w.flags, sizeof(int) * 3, hipMemcpyDeviceToHost, st));
  FLGP_HIP(stream_wait(st));
  if (sweeps_out) *sweeps_out = h_flags[2];
  if (strict && to_convergence && !h_flags[1]) {
    set_error("block Jacobi did not converge in %d sweeps (b=%d)", max_sweeps, b);
    return FLGP_ERR_NOCONV;
  }
  return FLGP_OK;
}

// symmetric eigendecomposition of the b x b matrix T (device): on return JV holds eigenvectors,
// h_lam the eigenvalues (unsorted, host copy).  Synchronises the stream.
static int jacobi_eig(hipStream_t st, const double *T, int ldt, int b, EigWork &w, std::vector<double> &h_lam,
                      int *sweeps_out, int sweep_limit = -1, double tol_scale = 1.0, bool strict = false,
                      bool to_host = true) {
  ProfScope ps("jacobi_eig", st, 8.0 * (double)b * b);
  hipLaunchKernelGGL(jac_init_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, T, ldt, b, w.JB, w.JV, b,
                     w.flags);
  FLGP_TRY(check_launch("jac_init_kernel"));
  return jacobi_run(st, b, w, h_lam, sweeps_out, sweep_limit, tol_scale, strict, nullptr, nullptr, to_host || sweep_limit < 0);
}

// Rayleigh-Ritz refinement for a T that is diagonal up to small couplings EXCEPT in its trailing
// g x g block (the guard columns never converge, so that block stays dense):
//   1. the g x g block is diagonalised completely by one workgroup inside LDS (one launch),
//   2. with V0 = blockdiag(I, Vg), B0 = T V0, `sweeps` global sweeps finish the job
//      (quadratic convergence: couplings eps -> eps^2 per sweep).
static int jacobi_refine(hipStream_t st, const double *T, int b, int K, EigWork &w, std::vector<double> &h_lam,
                         int *sweeps_out, int sweeps, bool to_host = true) {
  const int g = b - K;
  if (g < 2 || 2 * (size_t)g * g > (size_t)b * b)
    return jacobi_eig(st, T, b, b, w, h_lam, sweeps_out, -1, 1.0);
  ProfScope ps("jacobi_refine", st, 8.0 * (double)b * b);
  double *Vg = w.X2;
  if (g <= 64) {
    const size_t lds_small = sizeof(double) * (4 * 64 * 64 + 2 * 64) + sizeof(int) * 64;
    FLGP_HIP(hipFuncSetAttribute((const void *)small_sym_eig_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_small));
    const double tol_g = 4.0 * std::sqrt((double)g) * 1.1102230246251565e-16;
    hipLaunchKernelGGL(small_sym_eig_kernel<64>, dim3(1), dim3(1024), lds_small, st, T + (size_t)K * b + K, b, g, Vg,
                       tol_g, tuning("eig_guard_sweeps", 1));
    FLGP_TRY(check_launch("small_sym_eig_kernel"));
  } else {
    // a guard block too large for one workgroup's LDS: the block Jacobi on the g x g matrix itself
    // (its own B / V pair inside X2; a few sweeps are enough, the global sweeps below finish)
    double *Bg = w.X2 + (size_t)g * g;
    hipLaunchKernelGGL(jac_init_kernel, dim3(ceil_div((long)g * g, 256)), dim3(256), 0, st, T + (size_t)K * b + K, b, g,
                       Bg, Vg, g, w.flags);
    FLGP_TRY(check_launch("jac_init_kernel"));
    std::vector<double> tmp;
    FLGP_TRY(jacobi_run(st, g, w, tmp, nullptr, 4, 1.0, false, Bg, Vg, false));
  }
  hipLaunchKernelGGL(embed_block_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, Vg, g, K, b, w.JV);
  FLGP_TRY(check_launch("embed_block_kernel"));
  // JB = T JV  (wave-per-tile kernel: no workspace, so this may run beside the big GEMMs of another stream)
  if (b % 16 == 0) {
    SmallGemmPair pr;
    pr.g[0] = SmallGemm{T, w.JV, nullptr, w.JB, 1.0, 0.0};
    pr.g[1] = pr.g[0];
    hipLaunchKernelGGL(small_gemm_kernel, dim3(b / 16, b / 16, 1), dim3(64), 0, st, pr, b);
    FLGP_TRY(check_launch("small_gemm_kernel"));
  } else {
    FLGP_TRY(gemm_launch(st, b, b, b, 1.0, T, 1, b, w.JV, 1, b, 0.0, nullptr, 0, 0, w.JB, 1, b, w.gemm_ws, w.gemm_ws_elems,
                         0.0, nullptr));
  }
  return jacobi_run(st, b, w, h_lam, sweeps_out, sweeps, 1.0, false, nullptr, nullptr, to_host);
}

// W = JV(:, order), order = the eigenvalues in w.lam descending, all on the device: w.perm = order, w.theta = sorted values
static int sorted_basis_dev(hipStream_t st, int b, EigWork &w, int transposed = 0) {
  hipLaunchKernelGGL(ritz_sort_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, st, w.lam, b, w.perm, w.theta);
  hipLaunchKernelGGL(permute_scale_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.JV, b, b, w.perm,
                     (const double *)nullptr, (const double *)nullptr, b, w.W, b, transposed);
  return check_launch("ritz_sort_kernel");
}

// W = JV(:, order) * diag(scale), order = eigenvalues descending
static int sorted_basis(hipStream_t st, const std::vector<double> &lam, const std::vector<double> *scale, int b,
                        int ncols, EigWork &w, std::vector<int> &order, const double *d_rowscale = nullptr, int transposed = 0) {
  order.resize(b);
  for (int j = 0; j < b; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lam[x] > lam[y]; });
  FLGP_HIP(hipMemcpyAsync(w.perm, order.data(), sizeof(int) * ncols, hipMemcpyHostToDevice, st));
  if (scale) FLGP_HIP(hipMemcpyAsync(w.scale, scale->data(), sizeof(double) * ncols, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(permute_scale_kernel, dim3(ceil_div((long)b * ncols, 256)), dim3(256), 0, st, w.JV, b, b,
                     w.perm, scale ? w.scale : nullptr, d_rowscale, ncols, w.W, b, transposed);
  FLGP_TRY(check_launch("permute_scale_kernel"));
  FLGP_HIP(stream_wait(st));  // order / scale are host vectors that may die after return
  return FLGP_OK;
}

}  // namespace flgp

using namespace flgp;

extern "C" size_t flgp_dev_eig_workspace(int s, int K) {
  if (K < 0 || K > s) K = s;
  return eig_workspace_bytes(s, K);
}

static int eig_topk_impl(void *stream, const double *dG, int ldg, int s, int K, double tol, double *d_values, double *dV,
                         int ldv, void *d_work, size_t work_bytes, int *info);

extern "C" int flgp_dev_eig_topk(void *stream, const double *dG, int ldg, int s, int K, double tol,
                                 double *d_values, double *dV, int ldv, void *d_work, size_t work_bytes,
                                 int *info) {
  FLGP_REQUIRE(s >= 1 && ldg >= s && ldv >= s, "eig: bad shape");
  HostCtxLease lease;   // second stream + events + pinned read-back slots for this solve
  const int rc = eig_topk_impl(stream, dG, ldg, s, K, tol, d_values, dV, ldv, d_work, work_bytes, info);
  if (rc != FLGP_OK) {
    // An early return may leave work of this solve on either stream (the Lanczos steps on the side stream, copies into
    // the context's host slots): it must not outlive the workspace the caller is about to free, nor the lease.
    if (g_ctx && g_ctx->side) (void)hipStreamSynchronize(g_ctx->side);
    (void)hipStreamSynchronize((hipStream_t)stream);
  }
  return rc;
}

static int eig_topk_impl(void *stream, const double *dG, int ldg, int s, int K, double tol, double *d_values, double *dV,
                         int ldv, void *d_work, size_t work_bytes, int *info) {
  hipStream_t st = (hipStream_t)stream;
  if (K < 0) K = s;
  FLGP_REQUIRE(K >= 1 && K <= s, "eig: need 1 <= K <= s (K=%d, s=%d)", K, s);
  FLGP_REQUIRE(work_bytes >= eig_workspace_bytes(s, K), "eig: workspace too small");
  if (tol <= 0.0) tol = 5e-11;   // relative residual of every wanted pair (Spectra's own tolerance is 1e-10)
  const bool dense = eig_use_dense(s, K);
  const int b = dense ? s : eig_block_size(s, K);
  FLGP_REQUIRE(!dense || s <= 16384, "eig: the full decomposition (K == s, or K close to s) is built for s <= 16384");

  // carve the workspace
  EigWork w;
  char *p = (char *)d_work;
  auto take = [&](size_t bytes) { char *q = p; p += align_up(bytes); return q; };
  const size_t big = sizeof(double) * (size_t)s * b;
  if (!dense) {
    w.Q = (double *)take(big); w.Y = (double *)take(big); w.Yp = (double *)take(big); w.Z = (double *)take(big);
    w.Qold = (double *)take(big);
  } else { w.Q = w.Y = w.Yp = w.Z = w.Qold = nullptr; }
  const size_t small = sizeof(double) * (size_t)b * b;
  w.T = (double *)take(small); w.JB = (double *)take(small); w.JV = (double *)take(small); w.W = (double *)take(small);
  w.X2 = (double *)take(small); w.Id = (double *)take(small);
  w.lam = (double *)take(sizeof(double) * b); w.scale = (double *)take(sizeof(double) * b);
  w.res = (double *)take(sizeof(double) * 2 * b); w.theta = w.res + b; w.dinv = (double *)take(sizeof(double) * b);
  w.apriori = (double *)take(sizeof(double) * 2 * APRIORI_BLOCKS);
  w.perm = (int *)take(sizeof(int) * b); w.flags = (int *)take(sizeof(int) * 16);
  w.tickets = (int *)take(sizeof(int) * GEMM_MAX_TICKETS);
  FLGP_HIP(hipMemsetAsync(w.tickets, 0, sizeof(int) * GEMM_MAX_TICKETS, st));
  w.gemm_ws_elems = dense ? 0 : eig_gemm_ws_elems(s, b);
  w.gemm_ws = dense ? nullptr : (double *)take(sizeof(double) * w.gemm_ws_elems);
  w.red = nullptr; w.redcnt = nullptr;
  if (!dense) {
    const size_t nt2 = (size_t)ceil_div(b, 16) * ceil_div(b, 16);
    w.red = (double *)take(sizeof(double) * (nt2 + 2));
    w.redcnt = (int *)(w.red + nt2);
    FLGP_HIP(hipMemsetAsync(w.redcnt, 0, sizeof(int), st));
  }
  BsG bs;
  if (!dense && s >= std::max(1024, tuning("eig_bs_min_s", 1536)) && s <= 65536 && tuning("eig_blocksparse", 1)) {   // (65536: bsg_lists_kernel keeps one int per 64-anchor tile in LDS)
    bsg_carve(bs, p, s, b);
    if (g_ctx && g_ctx->pinned && tuning("eig_pinned_slots", 1))
      bsg_host_slots(bs, g_ctx->pinned, (char *)g_ctx->pinned + HOST_SMALL_BYTES, HOST_BIG_BYTES);
    FLGP_TRY(bsg_setup(st, dG, ldg, s, bs, g_ctx ? g_ctx->side : nullptr, g_ctx ? g_ctx->side_ev : nullptr));
  }

  std::vector<double> lam;
  std::vector<int> order;
  int sweeps = 0;
  if (info) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }

  if (dense) {
    // full symmetric eigendecomposition of G itself (the K == s branch, src/TruncatedSVD.cpp:17-20)
    FLGP_TRY(jacobi_eig(st, dG, ldg, s, w, lam, &sweeps, -1, 1.0, true));
    FLGP_TRY(sorted_basis(st, lam, nullptr, s, K, w, order));
    std::vector<double> vals(K);
    for (int j = 0; j < K; ++j) vals[j] = lam[order[j]];
    FLGP_HIP(hipMemcpyAsync(d_values, vals.data(), sizeof(double) * K, hipMemcpyHostToDevice, st));
    FLGP_HIP(hipMemcpy2DAsync(dV, sizeof(double) * ldv, w.W, sizeof(double) * s, sizeof(double) * s, K,
                              hipMemcpyDeviceToDevice, st));
    FLGP_HIP(stream_wait(st));
    if (info) { info[0] = sweeps; info[2] = 1; }
    return FLGP_OK;
  }

  const long tot = (long)s * b;
  auto gemmG = [&](const double *Xin, double alpha, double beta, const double *E, double gamma, const double *E2,
                   double *out) {  // out = alpha G Xin + beta E + gamma E2   (s x b)
    return gemm_launch(st, s, b, s, alpha, dG, 1, ldg, Xin, 1, s, beta, E, 1, s, out, 1, s, w.gemm_ws,
                       w.gemm_ws_elems, gamma, E2, w.tickets);
  };
  // ---- block-sparse products (bs.on): blocks live transposed (b x s, the b values of one row contiguous) while
  //      the filter runs, so that both the tiled GEMM (over the listed k stages of P G P^T) and the CSR remainder
  //      read and write whole 8b-byte rows
  auto to_t = [&](const double *in, double *out_t) {     // s x b  ->  b x s
    hipLaunchKernelGGL(bs_transpose_kernel, dim3(ceil_div(s, 32), ceil_div(b, 32)), dim3(256), 0, st, in, s, b, out_t);
    return check_launch("bs_transpose_kernel");
  };
  auto from_t = [&](const double *in_t, double *out) {   // b x s  ->  s x b
    hipLaunchKernelGGL(bs_transpose_kernel, dim3(ceil_div(b, 32), ceil_div(s, 32)), dim3(256), 0, st, in_t, b, s, out);
    return check_launch("bs_transpose_kernel");
  };
  auto gemmG_t = [&](const double *Xt, double alpha, double beta, const double *Et, double gamma, const double *E2t,
                     double *out_t) -> int {   // out_t = alpha X_t G' + beta E_t + gamma E2_t   (b x s)
    return bsg_product(st, bs, Xt, b, alpha, beta, Et, gamma, E2t, out_t);
  };
  const double *t_q = nullptr, *t_z = nullptr;   // blocks whose transposes currently sit in bs.T[0], bs.T[1]
  auto gram_small = [&](const double *Xa, const double *Xb, double *out, GemmFusedReduce *fr = nullptr) {  // out = Xa^T Xb   (b x b)
    if (gramk_applicable(s, b, Xa, Xb, w.gemm_ws_elems)) return gramk_launch(st, s, b, Xa, Xb, out, w.gemm_ws, w.gemm_ws_elems, fr);
    return gemm_launch(st, b, b, s, 1.0, Xa, s, 1, Xb, 1, s, 0.0, nullptr, 0, 0, out, 1, b, w.gemm_ws,
                       w.gemm_ws_elems, 0.0, nullptr, w.tickets, fr);
  };
  const bool fuse_reduce = tuning("eig_fused_reduce", 1) != 0;   // the small-matrix kernels behind a Gram product inside its reduction
  // The rotations run on their own kernel (rot.hip) where the shape allows; W is then kept k-major (`wt` = 1 tells its producers)
  const bool use_rot = rot_applicable(s, b, w.Q, w.Y, w.W, w.Yp, w.Z) && rot_applicable(s, b, w.Qold, nullptr, w.T, w.Z, nullptr);
  const int wt = use_rot ? 1 : 0;
  auto rotate = [&](const double *Xin, const double *Wm, double *out) {  // out = Xin Wm   (s x b)(b x b)
    if (use_rot) return rot_launch(st, s, b, 1.0, Xin, nullptr, Wm, 0.0, nullptr, nullptr, out, nullptr);
    return gemm_launch(st, s, b, b, 1.0, Xin, 1, s, Wm, 1, b, 0.0, nullptr, 0, 0, out, 1, s, w.gemm_ws,
                       w.gemm_ws_elems, 0.0, nullptr, w.tickets);
  };
  // two rotations by the same W in one launch (the Ritz vectors and G times them)
  const bool pair_rot = tuning("eig_pair_rotate", 1) != 0;
  auto rotate2 = [&](const double *X1, const double *X2, const double *Wm, double *out1, double *out2) -> int {
    if (!pair_rot) { FLGP_TRY(rotate(X1, Wm, out1)); return rotate(X2, Wm, out2); }
    if (use_rot) return rot_launch(st, s, b, 1.0, X1, X2, Wm, 0.0, nullptr, nullptr, out1, out2);
    const GemmPair pr{X2, Wm, out2};
    return gemm_launch(st, s, b, b, 1.0, X1, 1, s, Wm, 1, b, 0.0, nullptr, 0, 0, out1, 1, s, w.gemm_ws, w.gemm_ws_elems, 0.0,
                       nullptr, nullptr, nullptr, &pr);
  };
  // orthonormalise the columns of Yin into Qout ("SVQB" on the column-normalised block, so that
  // the widely different column norms a Chebyshev filter leaves behind do not enter the
  // conditioning of the Gram matrix); returns the condition estimate of the scaled Gram matrix
  const bool tiny_ok = b % 16 == 0 && tuning("eig_small_gemm", 1);
  auto small_gemm = [&](const double *Am, const double *Bm, double alpha, double beta, const double *E,
                        double *out) -> int {  // out = alpha Am Bm + beta E   (b x b, column-major)
    if (tiny_ok) {
      SmallGemmPair pr;
      pr.g[0] = SmallGemm{Am, Bm, E, out, alpha, beta};
      pr.g[1] = pr.g[0];
      ProfScope ps("small_gemm_kernel", st, 2.0 * (double)b * b * b);
      hipLaunchKernelGGL(small_gemm_kernel, dim3(b / 16, b / 16, 1), dim3(64), 0, st, pr, b);
      return check_launch("small_gemm_kernel");
    }
    return gemm_launch(st, b, b, b, alpha, Am, 1, b, Bm, 1, b, beta, E, 1, b, out, 1, b, w.gemm_ws, w.gemm_ws_elems,
                       0.0, nullptr);
  };
  // two independent products in one launch: out0 = A0 B0, out1 = A1 B1
  auto small_gemm2 = [&](const double *A0, const double *B0, double *out0, const double *A1, const double *B1,
                         double *out1) -> int {
    if (tiny_ok) {
      SmallGemmPair pr;
      pr.g[0] = SmallGemm{A0, B0, nullptr, out0, 1.0, 0.0};
      pr.g[1] = SmallGemm{A1, B1, nullptr, out1, 1.0, 0.0};
      ProfScope ps("small_gemm_kernel", st, 4.0 * (double)b * b * b);
      hipLaunchKernelGGL(small_gemm_kernel, dim3(b / 16, b / 16, 2), dim3(64), 0, st, pr, b);
      return check_launch("small_gemm_kernel");
    }
    FLGP_TRY(small_gemm(A0, B0, 1.0, 0.0, nullptr, out0));
    return small_gemm(A1, B1, 1.0, 0.0, nullptr, out1);
  };
  // host-visible result slots (HostCtx::pinned): [bsg | a-priori bounds | dist partials | residuals + Ritz values]
  static_assert(DIST_BLOCKS <= DIST_SLOT_DOUBLES, "slot of the distance partials");
  const bool host_slots = g_ctx && g_ctx->pinned && g_ctx->pinned_dev && tuning("eig_host_slots", 1);
  const size_t dist_off = BSG_HOST_SLOT_BYTES + sizeof(double) * APRIORI_SLOT_DOUBLES;
  const size_t rt_off = dist_off + sizeof(double) * DIST_SLOT_DOUBLES;
  double *dist_h = host_slots ? (double *)((char *)g_ctx->pinned + dist_off) : nullptr;
  double *dist_d = host_slots ? (double *)((char *)g_ctx->pinned_dev + dist_off) : nullptr;
  double *rt_h = (host_slots && 2 * b <= RT_SLOT_DOUBLES) ? (double *)((char *)g_ctx->pinned + rt_off) : nullptr;
  double *rt_d = rt_h ? (double *)((char *)g_ctx->pinned_dev + rt_off) : nullptr;
  auto dist_to_identity = [&](const double *M, double *out, double diag = 1.0) -> int {
    double part_own[DIST_BLOCKS];
    const double *part = part_own;
    if (dist_d) {
      hipLaunchKernelGGL(dist_to_identity_kernel, dim3(DIST_BLOCKS), dim3(256), 0, st, M, b, dist_d, diag);
      FLGP_TRY(check_launch("dist_to_identity_kernel"));
      part = dist_h;
    } else {
      hipLaunchKernelGGL(dist_to_identity_kernel, dim3(DIST_BLOCKS), dim3(256), 0, st, M, b, w.res, diag);
      FLGP_TRY(check_launch("dist_to_identity_kernel"));
      FLGP_HIP(hipMemcpyAsync(part_own, w.res, sizeof(double) * DIST_BLOCKS, hipMemcpyDeviceToHost, st));
    }
    FLGP_HIP(stream_wait(st));
    double sum = 0.0;
    for (int q = 0; q < DIST_BLOCKS; ++q) sum += part[q];
    *out = std::sqrt(sum);
    return FLGP_OK;
  };
  // two distances for one question to the host (the second lands in the upper half of the slot)
  static_assert(2 * DIST_BLOCKS <= DIST_SLOT_DOUBLES, "two sets of distance partials share the slot");
  auto dist_to_identity2 = [&](const double *M1, const double *M2, double *out1, double *out2) -> int {
    if (!dist_d) { FLGP_TRY(dist_to_identity(M1, out1)); return dist_to_identity(M2, out2); }
    hipLaunchKernelGGL(dist_to_identity_kernel, dim3(DIST_BLOCKS), dim3(256), 0, st, M1, b, dist_d);
    hipLaunchKernelGGL(dist_to_identity_kernel, dim3(DIST_BLOCKS), dim3(256), 0, st, M2, b, dist_d + DIST_BLOCKS);
    FLGP_TRY(check_launch("dist_to_identity_kernel"));
    FLGP_HIP(stream_wait(st));
    double s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < DIST_BLOCKS; ++q) { s1 += dist_h[q]; s2 += dist_h[DIST_BLOCKS + q]; }
    *out1 = std::sqrt(s1); *out2 = std::sqrt(s2);
    return FLGP_OK;
  };
  hipLaunchKernelGGL(set_identity_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.Id, b);
  FLGP_TRY(check_launch("set_identity_kernel"));
  int ns_orths = 0, jac_orths = 0;
  // the first step of the coupled iteration, where Z = I:  M = beta I + alpha Y (into Zout: it IS the next Z),  Yout = Y M
  const bool ns_first = tuning("eig_ns_first_step", 1) != 0;
  auto first_step = [&](const double *Yc_, double alpha, double beta, double *Zout, double *Yout) -> int {
    hipLaunchKernelGGL(eig_axpby_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, alpha, Yc_, beta, w.Id, Zout, (long)b * b);
    FLGP_TRY(check_launch("eig_axpby_kernel"));
    return small_gemm(Yc_, Zout, 1.0, 0.0, nullptr, Yout);
  };
  static_assert(GEMM_DIST_PARTS == DIST_BLOCKS, "the fused reduction answers in the slots of dist_to_identity_kernel");
  auto orth = [&](const double *Yin, double *Qout, double *cond_out) -> int {
    double delta = 0.0;
    // S = D Y^T Y D and |I - S|_F: inside the reduction kernel of the split product (one launch instead of four)
    GemmFusedReduce fr{1 | 4, w.dinv, dist_d ? dist_d : w.res, w.red, w.redcnt, false};
    FLGP_TRY(gram_small(Yin, Yin, w.T, fuse_reduce ? &fr : nullptr));
    if (fr.done) {
      double part_own[DIST_BLOCKS];
      const double *part = dist_d ? dist_h : part_own;
      if (!dist_d) FLGP_HIP(hipMemcpyAsync(part_own, w.res, sizeof(double) * DIST_BLOCKS, hipMemcpyDeviceToHost, st));
      FLGP_HIP(stream_wait(st));
      double sum = 0.0;
      for (int q = 0; q < DIST_BLOCKS; ++q) sum += part[q];
      delta = std::sqrt(sum);
    } else {
      hipLaunchKernelGGL(sym_scale_diag_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, st, w.T, b, w.dinv);
      hipLaunchKernelGGL(sym_scale_apply_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.T, b, w.dinv);
      FLGP_TRY(check_launch("sym_scale_kernel"));
      FLGP_TRY(dist_to_identity(w.T, &delta));
    }
    if (delta < 0.1 * tuning("eig_ns_plain_below_x10", 45)) {  // |I - S|_F bounds the spectral norm from above, loosely: try, and watch it contract
                                                              // (measured at configs[2]: 3.6 contracted, 5.7 did not)
      // well-conditioned block: S^-1/2 by the coupled Newton-Schulz iteration -- b x b MFMA GEMMs only
      //   M = (3 I - Z Y)/2,  Y <- Y M,  Z <- M Z ;  Y -> S^1/2, Z -> S^-1/2   (|I - S| < 1)
      double *Yc = w.T, *Zc = w.JV, *Mm = w.W, *Yn = w.JB, *Zn = w.X2;
      // (Z_0 = I is never stored: the first step is M = (3 I - Y)/2, Y <- Y M, Z <- M -- one product instead of three)
      if (!ns_first) FLGP_HIP(hipMemcpyAsync(Zc, w.Id, sizeof(double) * (size_t)b * b, hipMemcpyDeviceToDevice, st));
      // the error contracts quadratically, e <- (3/4) e^2 + O(e^3): run the predicted number of
      // iterations without talking to the host, then check once (|I - S|_F over-estimates e, so
      // the prediction errs on the safe side); a block that does not contract falls through to Jacobi
      // (|I - S|_F over-estimates the spectral norm that contracts: the prediction starts from 0.3 |I - S|_F -- over eight start
      //  blocks at configs[2] one to five steps fewer per orthonormalisation with the closing check still at rounding level,
      //  13.84 -> 13.70 ms; a prediction that falls short fails that check and the block takes the scaled iteration below)
      int kmax = tuning("eig_ns_extra", 2);   // steps beyond the predicted count (the closing check tests the LAST step's M, so one is inherent)
      for (double e = std::min(delta * 0.01 * tuning("eig_ns_e0_pct", 30), 0.95); e > 1e-17 && kmax < 40; ++kmax) e = (e < 0.5) ? 0.8 * e * e : 0.5 * e + 0.4 * e * e;
      bool ok = false;
      for (int k = 0; k < kmax; ++k) {
        if (k == 0 && ns_first) {
          FLGP_TRY(first_step(Yc, -0.5, 1.5, Zn, Yn));
          if (kmax == 1) FLGP_HIP(hipMemcpyAsync(Mm, Zn, sizeof(double) * (size_t)b * b, hipMemcpyDeviceToDevice, st));   // (the closing check reads Mm)
        } else {
          FLGP_TRY(small_gemm(Zc, Yc, -0.5, 1.5, w.Id, Mm));
          FLGP_TRY(small_gemm2(Yc, Mm, Yn, Mm, Zc, Zn));
        }
        std::swap(Yc, Yn);
        std::swap(Zc, Zn);
      }
      {
        // the last M must be the identity to rounding.  Its check is a host round trip: the rotation it would
        // allow is enqueued behind the measurement first (a failed check just overwrites Qout later), so the
        // GPU multiplies while the host reads the verdict
        double part_own[DIST_BLOCKS];
        const double *part = dist_d ? dist_h : part_own;
        hipLaunchKernelGGL(dist_to_identity_kernel, dim3(DIST_BLOCKS), dim3(256), 0, st, Mm, b, dist_d ? dist_d : w.res);
        FLGP_TRY(check_launch("dist_to_identity_kernel"));
        if (!dist_d) FLGP_HIP(hipMemcpyAsync(part_own, w.res, sizeof(double) * DIST_BLOCKS, hipMemcpyDeviceToHost, st));
        FLGP_HIP(stream_mark(st));
        // Zc may live in JV or X2; the rotation needs diag(dinv) Z in W (Mm's buffer: read by the kernel above first)
        hipLaunchKernelGGL(row_scale_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, Zc, b, w.dinv, w.W, 1.0, wt);
        FLGP_TRY(check_launch("row_scale_kernel"));
        FLGP_TRY(rotate(Yin, w.W, Qout));
        FLGP_HIP(mark_wait());
        double dm = 0.0;
        for (int q = 0; q < DIST_BLOCKS; ++q) dm += part[q];
        dm = std::sqrt(dm);
        ok = dm < 1e-13 * std::sqrt((double)b);
        if (tuning("eig_verbose", 0) > 1) fprintf(stderr, "[flgp orth] delta=%.3e kmax=%d dm=%.2e %s\n", delta, kmax, dm, ok ? "ok" : "FAILED");
      }
      if (ok) {
        if (cond_out) *cond_out = (1.0 + delta) / std::max(1.0 - delta, 1e-3);
        ++ns_orths;
        return FLGP_OK;
      }
      // did not contract: rebuild S and fall through to Jacobi
      FLGP_TRY(gram_small(Yin, Yin, w.T));
      hipLaunchKernelGGL(sym_scale_apply_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.T, b, w.dinv);
      FLGP_TRY(check_launch("sym_scale_kernel"));
    }
    if (tuning("eig_ns_scaled", 1)) {
      // ill-conditioned block (the first filters amplify by 1e3 and leave cond(S) ~ 1e5): the same
      // Newton-Schulz iteration on S / sigma with sigma >= lambda_max.  Every eigenvalue x of Z Y obeys
      // x <- x (3 - x)^2 / 4, which maps (0, 3) into (0, 1] and lifts a small x by 9/4 per step until
      // the quadratic phase: ~log(cond)/log(2.25) + 5 iterations of three b x b GEMMs, against twenty
      // Jacobi sweeps.  sigma = |S^2|_F^(1/2) over-estimates lambda_max by at most b^(1/4).
      double *Yc = w.T, *Zc = w.JV, *Mm = w.W, *Yn = w.JB, *Zn = w.X2;
      FLGP_TRY(small_gemm(Yc, Yc, 1.0, 0.0, nullptr, Yn));
      // (sigma computed on the device -- one question to the host less, one launch more -- measured: 14.90 ms either way)
      double f2 = 0.0;
      FLGP_TRY(dist_to_identity(Yn, &f2, 0.0));      // |S^2|_F
      const double sigma = 1.02 * std::sqrt(f2);
      if (sigma > 0.0 && std::isfinite(sigma)) {
        hipLaunchKernelGGL(eig_axpby_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, 1.0 / sigma, Yc, 0.0, w.Id,
                           Yn, (long)b * b);
        FLGP_TRY(check_launch("eig_axpby_kernel"));
        std::swap(Yc, Yn);
        bool z_is_identity = ns_first;      // Z_0 = I not stored until a step needs it (see first_step)
        if (!z_is_identity) FLGP_HIP(hipMemcpyAsync(Zc, w.Id, sizeof(double) * (size_t)b * b, hipMemcpyDeviceToDevice, st));
        double dm = 1.0;
        double zn = 0.0;   // |Z - I|_F^2 + 1 >= 1/x_min: an upper bound of cond(S / sigma)
        bool zn_valid = false;
        bool ok = false;
        // Dynamically scaled steps (Chen & Chow 2014): with every singular value x of the iterate in [l, 1],
        //   x <- (a / 2) x (3 - a^2 x^2),  a = sqrt(3 / (1 + l + l^2)),
        // is the cubic that lifts the lower end the most, l <- (a / 2) l (3 - a^2 l^2) (x 2.6 per step while l is small,
        // against 1.5 unscaled), and keeps [l, 1] inside itself.  l_0 is a guess: one that is too low costs a few steps,
        // one that is too high leaves singular values behind that the unscaled steps below then pick up at their own pace.
        // The step count follows from l alone, so the host is not asked until the end.
        int kdyn = 0;
        {
          double ell = std::pow(10.0, -(double)tuning("eig_ns_ell0_exp", 4));
          const int tail = tuning("eig_ns_tail", 2);
          int after = 0;
          while (kdyn < 60 && tuning("eig_ns_dynamic", 1)) {
            const bool plain = ell > 1.0 - 1e-9;
            if (plain && after++ >= tail) break;
            const double a = plain ? 1.0 : std::sqrt(3.0 / (1.0 + ell + ell * ell));
            if (z_is_identity) {
              FLGP_TRY(first_step(Yc, -0.5 * a * a * a, 1.5 * a, Zn, Yn));
              z_is_identity = false;
            } else {
              FLGP_TRY(small_gemm(Zc, Yc, -0.5 * a * a * a, 1.5 * a, w.Id, Mm));
              FLGP_TRY(small_gemm2(Yc, Mm, Yn, Mm, Zc, Zn));
            }
            std::swap(Yc, Yn);
            std::swap(Zc, Zn);
            ell = std::min(1.0, 0.5 * a * ell * (3.0 - a * a * ell * ell));
            ++kdyn;
          }
          if (kdyn == 1 && ns_first)   // (the only step was the first one: its M sits in Zc, not in Mm)
            FLGP_HIP(hipMemcpyAsync(Mm, Zc, sizeof(double) * (size_t)b * b, hipMemcpyDeviceToDevice, st));
          if (kdyn) {
            FLGP_TRY(dist_to_identity2(Mm, Zc, &dm, &zn));      // (zn is wanted only if dm passes: asked in the same breath)
            ok = std::isfinite(dm) && dm < 1e-9;
            zn_valid = ok;
          }
        }
        if (z_is_identity) {   // (no scaled step ran)
          FLGP_HIP(hipMemcpyAsync(Zc, w.Id, sizeof(double) * (size_t)b * b, hipMemcpyDeviceToDevice, st));
          z_is_identity = false;
        }
        for (int k = 0; k < 72 && !ok && std::isfinite(dm); ++k) {
          FLGP_TRY(small_gemm(Zc, Yc, -0.5, 1.5, w.Id, Mm));
          FLGP_TRY(small_gemm2(Yc, Mm, Yn, Mm, Zc, Zn));
          std::swap(Yc, Yn);
          std::swap(Zc, Zn);
          if ((kdyn || k >= 8) && k % 3 == 2) {
            FLGP_TRY(dist_to_identity(Mm, &dm));
            if (!std::isfinite(dm)) break;
            ok = dm < 1e-9;
          }
        }
        if (tuning("eig_verbose", 0) > 1) fprintf(stderr, "[flgp orth] scaled: delta=%.3e sigma=%.3e scaled steps=%d dm=%.2e %s\n", delta, sigma, kdyn, dm, ok ? "ok" : "FAILED");
        if (ok) {
          if (!zn_valid) FLGP_TRY(dist_to_identity(Zc, &zn));
          hipLaunchKernelGGL(row_scale_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, Zc, b, w.dinv, w.W,
                             1.0 / std::sqrt(sigma), wt);  // W = D (Z / sqrt(sigma))
          FLGP_TRY(check_launch("row_scale_kernel"));
          if (cond_out) {
            *cond_out = 1.0 + zn * zn;
            if (dm > 1e-12) *cond_out = std::max(*cond_out, 2e8);   // ask for the second pass
          }
          ++ns_orths;
          return rotate(Yin, w.W, Qout);
        }
      }
      FLGP_TRY(gram_small(Yin, Yin, w.T));
      hipLaunchKernelGGL(sym_scale_apply_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.T, b, w.dinv);
      FLGP_TRY(check_launch("sym_scale_kernel"));
    }
    ++jac_orths;
    FLGP_TRY(jacobi_eig(st, w.T, b, b, w, lam, &sweeps));
    double lmax = 0.0;
    for (int j = 0; j < b; ++j) lmax = std::max(lmax, lam[j]);
    std::vector<int> ord(b);
    for (int j = 0; j < b; ++j) ord[j] = j;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return lam[x] > lam[y]; });
    std::vector<double> sc(b);
    const double floor_ = lmax * 1e-28;
    double lmin = lmax;
    for (int j = 0; j < b; ++j) {
      const double l = std::max(lam[ord[j]], floor_);
      lmin = std::min(lmin, l);
      sc[j] = 1.0 / std::sqrt(l);
    }
    if (cond_out) *cond_out = (lmin > 0.0) ? lmax / lmin : 1e300;
    FLGP_TRY(sorted_basis(st, lam, &sc, b, b, w, order, w.dinv, wt));
    return rotate(Yin, w.W, Qout);
  };

  // ---- start block: uniform random columns.  NOT orthonormalised (round 3): iteration 0 forms no Rayleigh-Ritz
  //      matrix -- it filters the block as it is on a-priori bounds, and a filter is linear -- and the block that comes out
  //      of the filter is orthonormalised anyway; a uniform random s x b block has condition
  //      ~ (sqrt s + sqrt b) / (sqrt s - sqrt b) (1.6 at BASELINE configs[2]), so nothing is lost but the 0.33 ms of a
  //      Gram product, 24 Newton-Schulz launches and two host round trips.  (eig_start_orths = 1 / 2: the old behaviour.)
  //      Four s x b buffers rotate through the roles Q (current block) and three free ones.
  double *Q = w.Q, *F[3] = {w.Y, w.Yp, w.Z};
  const int start_orths = tuning("eig_start_orths", 1);   // (0 was tried in round 3: the 0.33 ms it saves sat in the shadow of the set-up's host round trips -- mean over eight start blocks 15.97 vs 15.83 ms)
  hipLaunchKernelGGL(eig_init_q_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, start_orths > 0 ? F[0] : Q, s, b, s,
                     (unsigned long long)tuning("eig_start_stream", 0));   // 0 = the documented start block; others: robustness runs
  FLGP_TRY(check_launch("eig_init_q_kernel"));
  // Rayleigh-Ritz on the random start block yields nothing but bounds, and poor ones (every Ritz value of a random
  // subspace sits near the mean eigenvalue); the span p(G) Q does not depend on the basis.  So the first filter runs
  // straight on the start block with a-priori bounds -- damped interval [0, trace / s], scaled at the 1-norm -- and the
  // first Rayleigh-Ritz step (two Jacobi sweeps, two rotations, a host round trip) is saved.
  const int skip_rr_n = tuning("eig_skip_rr0", 1) ? std::max(1, tuning("eig_skip_rr_n", 1)) : 0;   // iterations without Rayleigh-Ritz
  const bool skip_rr0 = skip_rr_n > 0;
  static_assert(2 * APRIORI_BLOCKS <= APRIORI_SLOT_DOUBLES, "the pinned slot of the a-priori bounds");
  double h_apriori_own[2 * APRIORI_BLOCKS];
  double *h_apriori = (g_ctx && g_ctx->pinned && tuning("eig_pinned_slots", 1)) ? (double *)((char *)g_ctx->pinned + BSG_HOST_SLOT_BYTES) : h_apriori_own;
  if (skip_rr0 && !bs.built) {
    hipLaunchKernelGGL(apriori_bounds_kernel, dim3(APRIORI_BLOCKS), dim3(256), 0, st, dG, ldg, s, w.apriori);
    FLGP_TRY(check_launch("apriori_bounds_kernel"));
    FLGP_HIP(hipMemcpyAsync(h_apriori, w.apriori, sizeof(double) * 2 * APRIORI_BLOCKS, hipMemcpyDeviceToHost, st));
  }
  double cond = 0.0;
  if (start_orths >= 2) {
    FLGP_TRY(orth(F[0], F[1], &cond));
    FLGP_TRY(orth(F[1], Q, &cond));
  } else if (start_orths == 1) {
    FLGP_TRY(orth(F[0], Q, &cond));
  } else if (!skip_rr0) {
    // an iteration 0 WITH Rayleigh-Ritz needs an orthonormal block
    FLGP_TRY(orth(Q, F[0], &cond));
    std::swap(Q, F[0]);
  }
  FLGP_HIP(stream_wait(st));   // the set-up's bookkeeping (and the a-priori bounds) have arrived on the host
  bsg_finish(bs);

  std::vector<double> theta(b), res(K), pred(K, 1.0);   // pred: see plan_filter (soft locking)
  bool soft_on = tuning("eig_soft_lock", 0) != 0;   // (off by default: see NOTEBOOK r04-1 -- 8 iterations instead of 10.4, but every one waits for two Jacobi sweeps)
  int last_m = 0;                                  // degree of the filter applied last
  int gprods = 0, it = 0;
  const int max_it = 80;
  bool converged = false;
  double *result = nullptr;
  double rmax_prev = 1.0;
  const int rr_every = soft_on ? 1 : tuning("eig_rr_every", 3);   // (soft locking needs every step's residuals; skipped steps were a saving of the 10-iteration schedule)
  int since_rr = 0, it_meas = 0;
  double rate = 0.1, rmax_meas = 0.0;

  // second stream: late Rayleigh-Ritz refinements (eight workgroups of Jacobi) run beside the filter's GEMMs
  struct Side {
    hipStream_t st;
    hipEvent_t ev;
  } side{g_ctx ? g_ctx->side : nullptr, g_ctx ? g_ctx->side_ev : nullptr};
  const bool can_overlap = side.st && side.ev && tuning("eig_overlap", 1);

  double lambda_lo = (bs.built && tuning("eig_lanczos_lo", 1)) ? bs.lambda_lo : 0.0;   // far end of the damped interval
  // ---- Chebyshev filter on [lo, cut], scaled to 1 at the top Ritz value
  struct FilterPlan { double c, e, sigma1; int m; };
  auto plan_filter = [&](double top, int it_) {
    const int cut_pos = K + (b - K) * tuning("eig_cut_pct", 90) / 100;
    double cut = theta[std::min(b - 1, std::max(K, cut_pos - 1))];
    if (!(cut > 0.0)) cut = 1e-3 * top;
    if (cut > 0.999 * top) cut = 0.999 * top;   // degenerate block: keep a valid interval
    FilterPlan fp;
    // damped interval [lo, cut]: lo = 0 (G is PSD) unless the set-up's Lanczos run vouches for more -- at BASELINE
    // configs[2] lambda_min = 0.113 and cut = 0.39: the interval shrinks by a quarter, the filter's growth per degree at the
    // K-th eigenvalue rises from 1.38 to 1.46, 17 % fewer products
    const double lo = (lambda_lo > 0.0 && lambda_lo < 0.5 * cut) ? lambda_lo : 0.0;
    fp.e = 0.5 * (cut - lo); fp.c = 0.5 * (cut + lo);
    // Soft locking (round 4).  The amplification cap exists for the columns that are still moving: a filtered column j
    // carries the rounding noise of the recurrence along the higher directions amplified by p(th_i) / p(th_j).  Along a
    // CONVERGED Ritz direction that noise is removed by the Gram-Schmidt pass below to the accuracy of the converged vector,
    // however large it was; so the cap is taken at the first Ritz value whose pair has NOT met the tolerance, not at the
    // top of the spectrum.  At BASELINE configs[2] the sixteen cluster eigenvalues (0.994..1, then a gap to 0.66) converge
    // by iteration 3 and the prefix grows by ~50 pairs per iteration: degrees 8 8 13 22 12 instead of 8 each, 7 outer
    // iterations instead of 10 with the same number of products (scripts/model_chfsi2.py soft=1; a cap of 1e9 still
    // converges in the model, 1e10 does not, and degrees beyond ~24 stop paying -- mmax).  The polynomial is still scaled to 1
    // at the true top (sigma1), so converged columns grow by up to T_m(g_top) ~ 1e23 per iteration and are renormalised by
    // the orthonormalisation's column scaling.  `pred` = the residuals of the last Rayleigh-Ritz step, contracted by the
    // filters applied since (overlapped iterations plan their filter before the step's own residuals are known).
    int n_soft = 0;
    if (it_ >= 2 && soft_on) {
      const double lim = tol * std::max(theta[0], 1e-300);
      while (n_soft < K - 1 && pred[n_soft] <= lim) ++n_soft;
    }
    const double top_act = std::max(theta[n_soft], 1e-300);
    const double g1 = (std::min(top, top_act) - fp.c) / fp.e;      // >= 1
    // degree: amplification T_m(g1) of the top direction capped per outer iteration
    // (gentler while the block is still far from the invariant subspace)
    const double amp = std::pow(10.0, (double)((it_ < 2) ? tuning("eig_amp_exp_early", 3) : tuning("eig_amp_exp", 8)));
    int m = (int)std::floor(std::acosh(amp) / std::acosh(std::max(g1, 1.0 + 1e-12)));
    const int m_cap = n_soft ? tuning("eig_soft_mmax", 24) : 40;
    fp.m = std::max(2, std::min(m, m_cap));
    // Landing.  A filter of degree m contracts the residual of the K-th pair -- the slowest -- by 1 / T_m(g_K) ~ 2 exp(-m a),
    // a = acosh(g_K), g_K the K-th Ritz value on the filter's own scale (measured at configs[2]: 0.0526 per iteration
    // against 1 / T_8(1.1055) = 0.0526).  With the last measured residual that gives the iterations still needed at the
    // capped degree, n0.  If a few degrees more per iteration save a whole iteration (orthonormalisation, Rayleigh-Ritz
    // and its wait: ~1 ms) they are spent; if n0 iterations overshoot, the degree is lowered to what the tolerance needs.
    // Without this the iteration on which the residual test is first met moves by one with perturbations of rounding
    // size (10 or 11 iterations at configs[2], depending on the start block).
    if (it_ >= 3 && tuning("eig_landing", 1) && rmax_prev > 0.0 && rmax_prev < 1e-8 * (double)tuning("eig_landing_below_e8", 100)) {
      const double gK = (theta[K - 1] - fp.c) / fp.e;
      if (gK > 1.0 + 1e-9) {
        const double a = std::acosh(gK), ln2 = 0.6931471805599453;
        const double target = tol * 0.01 * (double)tuning("eig_landing_margin_pct", 40);
        const double L = std::log(rmax_prev / target);
        const double per0 = fp.m * a - ln2;
        if (L > 0.0 && per0 > 0.0) {
          const int n0 = std::max(1, (int)std::ceil(L / per0));
          auto degree_for = [&](int n) { return (int)std::ceil((L / n + ln2) / a); };
          int mm = degree_for(n0);                                  // <= fp.m by construction of n0
          // (one iteration fewer at most, and three degrees more at most: with five or eight more per iteration the block
          //  loses accuracy faster than the filter gains -- 14 and 15 iterations instead of 10)
          if (n0 >= 2 && degree_for(n0 - 1) <= fp.m + tuning("eig_landing_boost", 3)) mm = degree_for(n0 - 1);
          fp.m = std::max(2, std::min(mm, m_cap));
          rate = std::min(0.5, std::max(1e-4, 2.0 * std::exp(-fp.m * a)));
        }
      }
    }
    fp.sigma1 = fp.e / (top - fp.c);
    if (tuning("eig_verbose", 0)) fprintf(stderr, "[flgp eig]   filter it=%d: soft prefix %d, active top %.4f, cut %.4f, lo %.3f, degree %d\n", it_, n_soft, top_act, cut, lo, fp.m);
    return fp;
  };
  // what a filter just applied does to the residual of pair j: 1 / T_m(g_j) ~ 2 exp(-m acosh g_j), times a safety factor
  // (the model: measured contraction within 3x of this for every pair once the block is past its first two iterations)
  auto contract_pred = [&](const FilterPlan &fp) {
    last_m = fp.m;
    const double safety = (double)tuning("eig_soft_safety", 10);
    for (int j = 0; j < K; ++j) {
      const double g = (theta[j] - fp.c) / fp.e;
      if (g > 1.0) pred[j] *= std::min(1.0, safety * 2.0 * std::exp(-fp.m * std::acosh(g)));
    }
  };
  // p(G) A given B = G A; A, f1, f2 are overwritten (B is not); returns the buffer with the result and
  // one buffer that is free afterwards
  auto apply_filter = [&](const FilterPlan &fp, double *A, const double *B, double *f1, double *f2, double **cur_out,
                          double **spare_out) -> int {
    double sigma = fp.sigma1;
    if (bs.on) {
      // the recurrence on transposed blocks; A and B are left alone, the result lands in f1
      if (!(t_q == A && t_z == B)) {
        FLGP_TRY(to_t(A, bs.T[0]));
        FLGP_TRY(to_t(B, bs.T[1]));
      }
      t_q = nullptr; t_z = nullptr;
      double *prev = bs.T[0], *cur = bs.T[2], *next = bs.T[1];
      hipLaunchKernelGGL(eig_axpby_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, fp.sigma1 / fp.e, bs.T[1],
                         -fp.sigma1 * fp.c / fp.e, bs.T[0], cur, tot);
      FLGP_TRY(check_launch("eig_axpby_kernel"));
      for (int deg = 2; deg <= fp.m; ++deg) {
        const double sn = 1.0 / (2.0 / fp.sigma1 - sigma);
        FLGP_TRY(gemmG_t(cur, 2.0 * sn / fp.e, -2.0 * sn * fp.c / fp.e, cur, -sigma * sn, prev, next));
        ++gprods;
        double *t3 = prev; prev = cur; cur = next; next = t3;
        sigma = sn;
      }
      FLGP_TRY(from_t(cur, f1));
      *cur_out = f1;
      *spare_out = f2;
      return FLGP_OK;
    }
    // degree 1: Y = (sigma1/e) (G A - c A) = (sigma1/e) (B - c A), into a free buffer
    double *prev = A, *cur = f1, *next = f2;
    hipLaunchKernelGGL(eig_axpby_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, fp.sigma1 / fp.e, B,
                       -fp.sigma1 * fp.c / fp.e, A, cur, tot);
    FLGP_TRY(check_launch("eig_axpby_kernel"));
    for (int deg = 2; deg <= fp.m; ++deg) {
      const double sn = 1.0 / (2.0 / fp.sigma1 - sigma);
      // next = (2 sn / e) (G cur - c cur) - sigma sn prev
      FLGP_TRY(gemmG(cur, 2.0 * sn / fp.e, -2.0 * sn * fp.c / fp.e, cur, -sigma * sn, prev, next));
      ++gprods;
      double *t3 = prev; prev = cur; cur = next; next = t3;
      sigma = sn;
    }
    *cur_out = cur;
    *spare_out = prev;
    return FLGP_OK;
  };
  // residuals of the K wanted pairs (A = Ritz vectors, B = G A) with the Ritz values sorted_basis_dev left in w.theta;
  // the ONE host round trip of a Rayleigh-Ritz step: brings the residuals and the sorted Ritz values over together
  std::vector<double> rt(2 * (size_t)b);
  auto residuals = [&](const double *A, const double *B, double *rmax_out) -> int {
    hipLaunchKernelGGL(resid_kernel, dim3(K), dim3(256), 0, st, B, A, s, s, w.theta, w.res, rt_d, b, 1);
    FLGP_TRY(check_launch("resid_kernel"));
    const double *rtp = rt_h;
    if (!rt_d) { FLGP_HIP(hipMemcpyAsync(rt.data(), w.res, sizeof(double) * 2 * b, hipMemcpyDeviceToHost, st)); rtp = rt.data(); }
    FLGP_HIP(stream_wait(st));
    double rmax = 0.0;
    for (int j = 0; j < K; ++j) { res[j] = rtp[j]; pred[j] = res[j]; rmax = std::max(rmax, res[j]); }
    for (int j = 0; j < b; ++j) theta[j] = rtp[b + j];
    *rmax_out = rmax;
    return FLGP_OK;
  };
  // Overlapped iterations plan their filter before the Rayleigh-Ritz step has finished.  The columns of Q are last
  // iteration's Ritz vectors, filtered, cleaned and orthonormalised -- nearly Ritz -- so | Z_j - Q_j T_jj | is a residual of
  // an approximate pair in its own right, and an upper estimate of the Ritz pair's: a MEASURED prefix of converged pairs for
  // the soft locking, one small kernel and one host round trip (~25 us) behind the Gram product that is waited for anyway.
  auto residual_estimate = [&](const double *Qc, const double *Zc, const double *Tm) -> int {
    hipLaunchKernelGGL(resid_kernel, dim3(K), dim3(256), 0, st, Zc, Qc, s, s, Tm, w.res, rt_d, b, b + 1);
    FLGP_TRY(check_launch("resid_kernel"));
    const double *rtp = rt_h;
    if (!rt_d) { FLGP_HIP(hipMemcpyAsync(rt.data(), w.res, sizeof(double) * K, hipMemcpyDeviceToHost, st)); rtp = rt.data(); }
    FLGP_HIP(stream_wait(st));
    for (int j = 0; j < K; ++j) pred[j] = rtp[j];
    return FLGP_OK;
  };
  auto after_rr = [&](double rmax, double top, bool overlapped) {   // book-keeping shared by both orders
    // the watch on the Lanczos bound: a direction below `lambda_lo` that the filter amplified instead of damping shows
    // up as a Ritz value far below the guard block's (which sit just under the K-th); then the bound goes
    if (lambda_lo > 0.0 && theta[b - 1] < lambda_lo + 0.5 * (theta[K - 1] - lambda_lo)) {
      if (tuning("eig_verbose", 0)) fprintf(stderr, "[flgp eig] smallest Ritz value %.4g: the lower bound %.4g is dropped\n", theta[b - 1], lambda_lo);
      lambda_lo = 0.0;
    }
    if (tuning("eig_verbose", 0)) {
      int npre = 0, nconv = 0;
      while (npre < K && res[npre] <= tol * top) ++npre;
      for (int j = 0; j < K; ++j) nconv += res[j] <= tol * top;
      fprintf(stderr, "[flgp eig] it=%d gprods=%d theta0=%.15g thetaK=%.6g cut=%.6g rmax=%.3e cond=%.2e sweeps=%d conv=%d prefix=%d%s\n",
              it, gprods, theta[0], theta[K - 1], theta[b - 1], rmax, cond, sweeps, nconv, npre, overlapped ? " (overlapped)" : "");
      if (tuning("eig_verbose", 0) > 2) {
        int shown = 0;
        for (int j = 0; j < K && shown < 8; ++j) if (res[j] > tol * top) { fprintf(stderr, "    pair %d theta %.9f res %.2e\n", j, theta[j], res[j]); ++shown; }
      }
    }
    if (rmax <= tol * top) return true;
    // the soft locking's safeguard: residuals that GROW say a filter was stronger than the block could take (a pair counted
    // as converged was not) -- back to the cap at the top of the spectrum for the rest of the solve
    if (soft_on && it >= 3 && rmax_meas > 0.0 && rmax / top > 4.0 * rmax_meas) {
      soft_on = false;
      if (tuning("eig_verbose", 0)) fprintf(stderr, "[flgp eig] residual grew (%.2e -> %.2e): soft locking off\n", rmax_meas, rmax / top);
    }
    if (it >= 3 && rmax_meas > 0.0 && rmax / top < rmax_meas) {
      const double rt = std::pow((rmax / top) / rmax_meas, 1.0 / (double)(it - it_meas));
      rate = std::min(0.5, std::max(0.02, rt));
    }
    rmax_meas = rmax / top; it_meas = it;
    rmax_prev = rmax / top;
    return false;
  };

  for (it = 0; it < max_it; ++it) {
    double *Z = F[0];
    double *A, *B, *free1, *free2;     // Ritz vectors, G * Ritz vectors, two free s x b buffers
    double *cur = nullptr, *spare = nullptr;   // filtered block; a buffer that is free after the filter
    if (bs.on) {
      FLGP_TRY(to_t(Q, bs.T[0]));
      FLGP_TRY(gemmG_t(bs.T[0], 1.0, 0.0, nullptr, 0.0, nullptr, bs.T[1]));
      FLGP_TRY(from_t(bs.T[1], Z));
      t_q = Q; t_z = Z;
    } else {
      FLGP_TRY(gemmG(Q, 1.0, 0.0, nullptr, 0.0, nullptr, Z));
    }
    ++gprods;
    // Rayleigh-Ritz may be skipped on some late iterations (rr_every > 1): the block is then used as it
    // is (its columns are the previous Ritz vectors, filtered, cleaned and orthonormalised: still ordered
    // and nearly Ritz), bounds are reused, and no convergence test is made on that iteration
    // (rmax_prev is then advanced by the measured per-iteration contraction `rate`, so that the Rayleigh-
    //  Ritz step and its convergence test land on the iteration where the tolerance is expected to be met)
    const bool near_done = rmax_prev * rate <= 4.0 * tol;
    const bool early_skip = (tuning("eig_skip_it1", 0) && it == 1) || it < skip_rr_n;
    const bool do_rr = !early_skip && !(rr_every > 1 && it >= 3 && rmax_prev < 1e-6 * tuning("eig_rr_skip_below_e6", 1000) && since_rr + 1 < rr_every && !near_done);
    // Late Rayleigh-Ritz steps only refine a nearly diagonal T: the filter does not wait for them.  It is
    // linear, p(G) (Q W) = (p(G) Q) W, so it runs on the block as it is, with the bounds of the previous
    // step (once the residuals are below 1e-3 they move in the third digit: measured, earlier steps lose
    // more to the stale interval than they gain), while the refinement runs on the second stream; the
    // rotation W is applied to the filtered block afterwards.  Not on the step that is expected to
    // converge: there the filter's products would be thrown away.
    const bool overlap = can_overlap && do_rr && it >= tuning("eig_overlap_from_it", 3) && !near_done &&
                         rmax_prev <= 1e-6 * (double)tuning("eig_overlap_below_e6", 10000);
    double rmax = rmax_prev * rate, top = std::max(theta[0], 1e-300);
    if (do_rr && overlap) {
      since_rr = 0;
      FLGP_TRY(gram_small(Q, Z, w.T));
      FLGP_HIP(hipEventRecord(side.ev, st));
      // the other stream first (round 4: it used to be enqueued BEHIND the filter's launches, and the host needs ~0.4 ms to
      // enqueue a filter of degree 24 -- the refinement, which is the longer of the two, started that much late): T = W Th W^T
      FLGP_HIP(hipStreamWaitEvent(side.st, side.ev, 0));
      {
        int nsw = std::max(1, (rmax_prev > 1e-4 * tuning("eig_refine3_above_e4", 2000) ? 3 : (rmax_prev > 1e-8 * tuning("eig_refine2_above_e8", 100) ? 2 : 1)) - tuning("eig_refine_minus", 0));
        if (last_m >= tuning("eig_sweeps2_from_m", 11)) nsw = std::max(nsw, 2);   // a strong filter leaves T further from diagonal than the residual says (measured: one sweep after degree 24 un-converged the cluster pairs)
        FLGP_TRY(jacobi_refine(side.st, w.T, b, K, w, lam, &sweeps, nsw, false));
        sweeps = nsw;
      }
      FLGP_TRY(sorted_basis_dev(side.st, b, w, wt));   // order and W on the device: the host is not asked
      FLGP_HIP(hipEventRecord(side.ev, side.st));
      FLGP_HIP(hipMemcpyAsync(w.Qold, Q, sizeof(double) * (size_t)tot, hipMemcpyDeviceToDevice, st));
      if (soft_on && tuning("eig_soft_estimate", 1)) FLGP_TRY(residual_estimate(Q, Z, w.T));
      const FilterPlan fp = plan_filter(top, it);
      FLGP_TRY(apply_filter(fp, Q, Z, F[1], F[2], &cur, &spare));
      FLGP_HIP(hipStreamWaitEvent(st, side.ev, 0));
      // the two buffers of {Q, F1, F2} that do not hold the filtered block take A and B
      double *trio[3] = {Q, F[1], F[2]};
      double *x[2]; int nx = 0;
      for (int q = 0; q < 3; ++q) if (trio[q] != cur) x[nx++] = trio[q];
      A = x[0]; B = x[1];
      FLGP_TRY(rotate2(w.Qold, Z, w.W, A, B));   // A = Ritz vectors, B = G * Ritz vectors
      FLGP_TRY(residuals(A, B, &rmax));
      top = std::max(theta[0], 1e-300);
      if (after_rr(rmax, top, true)) { converged = true; result = A; break; }
      contract_pred(fp);                  // (the residuals just measured are those of the block BEFORE this iteration's filter)
      FLGP_TRY(rotate(cur, w.W, Z));      // the filtered block in the new Ritz order (Z is free by now)
      FLGP_HIP(hipMemcpyAsync(w.Qold, A, sizeof(double) * (size_t)tot, hipMemcpyDeviceToDevice, st));
      free1 = cur; free2 = Z;
      spare = cur;
      cur = Z;
    } else {
      if (do_rr) {
        since_rr = 0;
        A = F[1]; B = F[2]; free1 = Q; free2 = Z;
        // ---- Rayleigh-Ritz on span(Q): Z = G Q, T = Q^T Z, T = W Th W^T
        FLGP_TRY(gram_small(Q, Z, w.T));
        // T is far from diagonal only while the block is far from invariant: full Jacobi for the first
        // iterations, afterwards a single sweep refines the (already nearly diagonal) Ritz basis
        // (a fixed small number of global sweeps alone is NOT enough, even late: the guard columns
        //  never converge, so their diagonal block of T stays dense -- measured: 2 sweeps put rmax
        //  back to 4e-2.  jacobi_refine diagonalises that block first.)
        if (it < 2) {       // bounds and a rough Ritz basis are all that is needed yet: loose threshold, capped sweeps
          sweeps = tuning(it == 0 ? "eig_sweeps_it0" : "eig_sweeps_it1", 2);
          FLGP_TRY(jacobi_eig(st, w.T, b, b, w, lam, nullptr, sweeps, 1e10, false, false));
        } else if (rmax_prev > 5e-2) {
          sweeps = tuning("eig_sweeps_it2", 2);
          FLGP_TRY(jacobi_eig(st, w.T, b, b, w, lam, nullptr, sweeps, 1e6, false, false));
        } else {
          sweeps = std::max(1, (rmax_prev > 1e-4 * tuning("eig_refine3_above_e4", 2000) ? 3 : (rmax_prev > 1e-8 * tuning("eig_refine2_above_e8", 100) ? 2 : 1)) - tuning("eig_refine_minus", 0));
          if (last_m >= tuning("eig_sweeps2_from_m", 11)) sweeps = std::max(sweeps, 2);
          FLGP_TRY(jacobi_refine(st, w.T, b, K, w, lam, nullptr, sweeps, false));
        }
        FLGP_TRY(sorted_basis_dev(st, b, w, wt));
        FLGP_TRY(rotate2(Q, Z, w.W, A, B));   // A = Ritz vectors, B = G * Ritz vectors
        FLGP_TRY(residuals(A, B, &rmax));
        top = std::max(theta[0], 1e-300);
        if (after_rr(rmax, top, false)) { converged = true; result = A; break; }
      } else {
        ++since_rr;
        A = Q; B = Z; free1 = F[1]; free2 = F[2];
        if (it >= skip_rr_n) rmax_prev *= rate;
      }
      FilterPlan fp;
      if (it < skip_rr_n) {   // (the start orthonormalisation has synchronised the stream: h_apriori is valid)
        double n1 = 0.0, tr = 0.0;
        if (bs.built) { n1 = bs.h_bounds[0]; tr = bs.h_bounds[1]; }   // gathered by the block-sparse set-up's pass over G
        else for (int q = 0; q < APRIORI_BLOCKS; ++q) { n1 = std::max(n1, h_apriori[q]); tr += h_apriori[APRIORI_BLOCKS + q]; }
        double cut0 = tr / (double)s;
        if (!(n1 > 0.0) || !std::isfinite(n1)) { set_error("eigensolver: the matrix is zero or not finite"); return FLGP_ERR_INVALID; }
        if (!(cut0 > 1e-3 * n1)) cut0 = 1e-3 * n1;
        if (cut0 > 0.5 * n1) cut0 = 0.5 * n1;
        fp.c = fp.e = 0.5 * cut0;
        fp.sigma1 = fp.e / (n1 - fp.c);
        fp.m = std::max(2, it == 0 ? tuning("eig_m0", 8) : tuning("eig_m1", 6));
        top = n1;
      } else {
        fp = plan_filter(top, it);
      }
      // the Ritz vectors are needed again after the filter (see below): keep a copy
      FLGP_HIP(hipMemcpyAsync(w.Qold, A, sizeof(double) * (size_t)tot, hipMemcpyDeviceToDevice, st));
      FLGP_TRY(apply_filter(fp, A, B, free1, free2, &cur, &spare));
      if (it >= skip_rr_n) contract_pred(fp);
    }
    // ---- de-contaminate: a filtered column y_j = p(G) q_j carries its error components along the
    //      higher Ritz directions amplified by p(th_i)/p(th_j) (up to `amp`).  One Gram-Schmidt pass
    //      against the OLD Ritz vectors in sorted order removes exactly those:
    //          y_j <- y_j - sum_{i<j} q_i (q_i . y_j)
    //      (two GEMMs).  What is left is nearly orthogonal, so the symmetric orthonormalisation below
    //      no longer mixes eigen-directions and the next T stays diagonal up to the guard block.
    //      (Not after the a-priori filter of iteration 0: the start block is no Ritz basis, and projecting along its
    //      columns would take the block out of span p(G) Q.)
    if (!(it < skip_rr_n)) {
      if (use_rot) {
        // T^T = cur^T Qold, strictly lower: the same sums (products commute), laid out k-major as rot.hip reads W
        GemmFusedReduce fr{8, nullptr, nullptr, w.red, w.redcnt, false};
        FLGP_TRY(gram_small(cur, w.Qold, w.T, fuse_reduce ? &fr : nullptr));
        if (!fr.done) {
          hipLaunchKernelGGL(mask_strict_upper_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.T, b, 1);
          FLGP_TRY(check_launch("mask_strict_upper_kernel"));
        }
        FLGP_TRY(rot_launch(st, s, b, -1.0, w.Qold, nullptr, w.T, 1.0, cur, nullptr, cur, nullptr));
      } else {
      GemmFusedReduce fr{2, nullptr, nullptr, w.red, w.redcnt, false};   // the strict upper triangle, by the reduction kernel
      FLGP_TRY(gram_small(w.Qold, cur, w.T, fuse_reduce ? &fr : nullptr));
      if (!fr.done) {
        hipLaunchKernelGGL(mask_strict_upper_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, w.T, b);
        FLGP_TRY(check_launch("mask_strict_upper_kernel"));
      }
      FLGP_TRY(gemm_launch(st, s, b, b, -1.0, w.Qold, 1, s, w.T, 1, b, 1.0, cur, 1, s, cur, 1, s, w.gemm_ws, w.gemm_ws_elems,
                           0.0, nullptr, w.tickets));
      }
    }
    // ---- orthonormalise the filtered block (B is free by now; twice if ill-conditioned)
    FLGP_TRY(orth(cur, B, &cond));
    double *R = B;
    if (cond > 1e8) {
      FLGP_TRY(orth(B, spare, &cond));
      R = spare;
    }
    // new roles: Q = R, the other three buffers are free
    double *pool[4] = {A, B, free1, free2};
    int nf = 0;
    for (int q = 0; q < 4; ++q)
      if (pool[q] != R) F[nf++] = pool[q];
    Q = R;
  }
  if (info) { info[0] = it; info[1] = gprods; info[2] = 0; info[3] = ns_orths * 1000 + jac_orths; }
  if (!converged) {
    // A spectrum the filter cannot split -- e.g. r = 1, where G is the identity up to the 1e-9 guards and the wanted
    // and unwanted eigenvalues coincide -- never meets the residual test.  While the full decomposition is affordable
    // (s <= 4096) it is taken instead, with its own workspace: any orthonormal basis of a degenerate eigenspace is a
    // valid answer, and the Jacobi route always delivers one.
    if (s <= 8192 && tuning("eig_dense_fallback", 1)) {
      DevBuf fw, fvals, fV;
      const size_t fb = eig_workspace_bytes(s, s);
      FLGP_TRY(fw.alloc(fb));
      FLGP_TRY(fvals.alloc(sizeof(double) * (size_t)s));
      FLGP_TRY(fV.alloc(sizeof(double) * (size_t)s * s));
      if (tuning("eig_verbose", 0)) fprintf(stderr, "[flgp eig] no convergence after %d outer iterations: full decomposition instead\n", max_it);
      FLGP_TRY(flgp_dev_eig_topk(stream, dG, ldg, s, s, tol, fvals.as<double>(), fV.as<double>(), s, fw.p, fb, nullptr));
      FLGP_HIP(hipMemcpyAsync(d_values, fvals.p, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
      FLGP_HIP(hipMemcpy2DAsync(dV, sizeof(double) * (size_t)ldv, fV.p, sizeof(double) * (size_t)s, sizeof(double) * (size_t)s, K,
                                hipMemcpyDeviceToDevice, st));
      FLGP_HIP(stream_wait(st));
      if (info) info[2] = 1;
      return FLGP_OK;
    }
    set_error("eigensolver: %d of the residuals still above %.1e after %d outer iterations", K, tol, max_it);
    return FLGP_ERR_NOCONV;
  }
  if (rt_h) {       // (through the pinned slot: an upload from pageable memory is staged by the runtime first)
    memcpy(rt_h, theta.data(), sizeof(double) * K);
    FLGP_HIP(hipMemcpyAsync(d_values, rt_h, sizeof(double) * K, hipMemcpyHostToDevice, st));
  } else {
    FLGP_HIP(hipMemcpyAsync(d_values, theta.data(), sizeof(double) * K, hipMemcpyHostToDevice, st));
  }
  if (bs.on) {   // the solver worked on P G P^T: rows back to the caller's anchor order
    hipLaunchKernelGGL(bs_unpermute_kernel, dim3(ceil_div((long)s * K, 256)), dim3(256), 0, st, result, s, K, bs.perm, dV, ldv);
    FLGP_TRY(check_launch("bs_unpermute_kernel"));
  } else {
    FLGP_HIP(hipMemcpy2DAsync(dV, sizeof(double) * ldv, result, sizeof(double) * s, sizeof(double) * s, K,
                              hipMemcpyDeviceToDevice, st));
  }
  FLGP_HIP(stream_wait(st));
  return FLGP_OK;
}

extern "C" void flgp_dev_jac_set_trace(void *d_trace) { flgp::g_jac_trace = (long long *)d_trace; }
