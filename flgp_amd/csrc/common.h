// Shared host-side helpers for libflgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/flgp_hip.h"

namespace flgp {

void set_error(const char *fmt, ...);
int tuning(const char *key, int dflt);

// optional per-launch HIP-event timing (core.hip); work = algorithmic flops or bytes of the launch
int prof_begin(const char *name, hipStream_t st, double work);
void prof_end(int idx, hipStream_t st);
struct ProfScope {
  int idx; hipStream_t st;
  ProfScope(const char *name, hipStream_t s, double work) : idx(prof_begin(name, s, work)), st(s) {}
  ~ProfScope() { prof_end(idx, st); }
};

inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %s (%d) in %s at %s:%d", hipGetErrorString(e), (int)e, what, file, line);
  return FLGP_ERR_HIP;
}

#define FLGP_HIP(call)                                                         \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return ::flgp::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define FLGP_TRY(call)            \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != FLGP_OK) return rc_; \
  } while (0)

#define FLGP_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      ::flgp::set_error(__VA_ARGS__); \
      return FLGP_ERR_INVALID;       \
    }                                \
  } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Device memory of the entry points comes from a small cache (core.hip): hipMalloc / hipFree cost 50-500 us each and an
// entry point makes dozens (2.7 ms of a 34 ms call at BASELINE configs[2]); a block that is given back is kept, up to
// `pool_max_mb` (default 16 GB) per process, and handed to the next request of the same size on the same device.  Giving
// back synchronises the device first -- exactly what the hipFree it replaces did -- so nothing in flight can still use
// the block when its next owner writes to it.  Blocks above 4 GB bypass the cache.  flgp_dev_pool_release() empties it.
void *pool_take(int dev, size_t bytes);                 // nullptr: nothing of that size cached
bool pool_give(int dev, void *p, size_t bytes);         // false: not kept (the caller frees)

// RAII device buffer for the host-pointer entry points
struct DevBuf {
  void *p = nullptr;
  bool owned = true;
  size_t cap = 0;       // bytes as allocated (rounded), for the cache
  int dev = -1;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) {
      bool kept = false;
      if (cap && dev >= 0) {
        // the hipFree this replaces waited for the OWNING device; the caller may be on another one by now (a rank's thread of
        // the multi-GPU entry, a flgp_set_device in between)
        int cur = -1;
        const bool here = hipGetDevice(&cur) == hipSuccess && cur == dev;
        if (here || hipSetDevice(dev) == hipSuccess) {
          (void)hipDeviceSynchronize();
          kept = pool_give(dev, p, cap);
          if (!here && cur >= 0) (void)hipSetDevice(cur);
        }
      }
      if (!kept) (void)hipFree(p);
    }
    p = nullptr; cap = 0; dev = -1;
  }
  void borrow(const void *q) { release(); p = (void *)q; owned = false; }   // the caller's device memory: never freed here
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    release();
    owned = true;
    bytes = (bytes + 255) / 256 * 256;
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && bytes <= ((size_t)4 << 30)) {
      p = pool_take(d, bytes);
      if (p) { cap = bytes; dev = d; return FLGP_OK; }
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {       // the memory may be parked in the cache under other sizes: empty it and ask once more
      (void)hipGetLastError();
      if (flgp_dev_pool_release() > 0) e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { p = nullptr; set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return FLGP_ERR_NOMEM; }
    if (bytes <= ((size_t)4 << 30)) { cap = bytes; dev = d; }
    return FLGP_OK;
  }
  template <class T> T *as() { return (T *)p; }
};

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return FLGP_ERR_HIP; }
  return FLGP_OK;
}

// C(i,j) = alpha * sum_k A(i,k) B(k,j) + beta * E(i,j) + gamma * E2(i,j)   (gemm.hip)
int gemm_launch(hipStream_t st, int M, int N, int Kd, double alpha, const double *A, long a_is, long a_ks,
                const double *B, long b_ks, long b_js, double beta, const double *E, long e_is, long e_js,
                double *C, long c_is, long c_js, double *work, size_t work_elems, double gamma,
                const double *E2, int *tickets = nullptr, struct GemmFusedReduce *fused = nullptr,
                const struct GemmPair *pair = nullptr);
// `pair`: a second product C2 = alpha A2 B2 of the same shape and strides in the same launch (no E / E2, never split):
// two of the solver's s x b rotations fill the chip where one leaves its fixed costs exposed.
struct GemmPair { const double *A2, *B2; double *C2; };
int gemm_reduce_square(hipStream_t st, int b, const double *part, int nsplit, double *C, GemmFusedReduce *fused);
// The eigensolver's b x b Gram product out = Xa^T Xb (Xa, Xb: s x b column-major) on its own kernel (rot.hip): split over the
// rows, planes reduced by gemm.hip's reduction kernels (with the fused extras).  false: not this shape, nothing was launched.
bool gramk_applicable(int s, int b, const double *Xa, const double *Xb, size_t work_elems);
int gramk_launch(hipStream_t st, int s, int b, const double *Xa, const double *Xb, double *out, double *work, size_t work_elems,
                 GemmFusedReduce *fused);
// The eigensolver's rotation out = alpha X W + beta E on its own kernel (rot.hip): W k-major, WT[k * b + j] = W(k, j)
bool rot_applicable(int s, int b, const double *X, const double *X2, const double *WT, const double *out, const double *out2);
int rot_launch(hipStream_t st, int s, int b, double alpha, const double *X, const double *X2, const double *WT, double beta,
               const double *E, const double *E2, double *out, double *out2);
// Optional extra work for the split-K reduction kernel of a SQUARE product S (M == N, alpha = 1, no E / E2), so that the
// eigensolver's small matrices need no kernels of their own behind the product:
//   mode bit 0: S <- D S D with D = diag(1 / sqrt(S_jj)) (0 where S_jj <= 0), D stored in dinv;
//        bit 1: only the strictly upper triangle (row < column) of S is kept, the rest zero (bit 3: the strictly lower one);
//        bit 2: |S - I|_F^2 (after bits 0 / 1) as GEMM_DIST_PARTS partial sums in a fixed order into dist[] (may be host memory).
// `scratch` holds one double per 16 x 16 tile of S, `counter` one int that is zero between launches.  `done` tells the caller
// whether the reduction kernel ran (the product was split) -- if not, S is the plain product and the caller runs its own kernels.
constexpr int GEMM_DIST_PARTS = 32;
struct GemmFusedReduce {
  int mode;
  double *dinv, *dist, *scratch;
  int *counter;
  bool done;
};
// `tickets`: GEMM_MAX_TICKETS zero-initialised ints owned by the caller (one stream at a time, like `work`); with them
// a split-K product finishes inside the GEMM kernel (the last piece of a tile to arrive adds the partial planes in a
// fixed order) instead of in a second launch.  The kernel leaves them zero.
constexpr int GEMM_MAX_TICKETS = 1024;


// heat-kernel contraction on LDS-resident panels of V (hk.hip); d_vw holds hk_panel_vw_elems(n1, K) doubles
bool hk_panel_applicable(int n0, int n1, int K, long ldh);
bool hk_panel2_applicable(int n0, int n1, int K, long ldh);    // hk2.hip: the same with the k loop unrolled (K in 97..112, 193..208)
size_t hk_panel_vw_elems(int n1, int K);
int hk_panel2_launch(hipStream_t st, const double *d_values, int K, double t, const double *V0, long ld0, int n0,
                     const double *dV1, int ld1, const int *d_idx1, int row0_1, int n1, double *dH, long ldh,
                     double *d_vw);
int hk_panel_launch(hipStream_t st, const double *d_values, int K, double t, const double *V0, long ld0, int n0,
                    const double *dV1, int ld1, const int *d_idx1, int row0_1, int n1, double *dH, long ldh,
                    double *d_vw);

// dense algebra of the regression consumers of an EigenPair (gpr.hip)
int chol_solve(hipStream_t st, double *dA, int N, double *dB, int nrhs, int *d_flag);
int gpr_weights(hipStream_t st, const double *d_values, int K, double t, double *d_ls, double *d_l);
int gpr_q(hipStream_t st, const double *dVtV, const double *d_ls, int K, double c, double *dQ);
int gpr_scale(hipStream_t st, const double *dM, const double *d_a, const double *d_b, int rows, int cols, double *d_out);
int gpr_diff(hipStream_t st, const double *dX, const double *dY, double alpha, long count, double *d_out);
int gpr_add_diag(hipStream_t st, double *dA, int N, double c);
int gpr_add_diag_vec(hipStream_t st, double *dA, int N, const double *d_v);
int gpr_rowscale_ld(hipStream_t st, const double *dM, long ldm, const double *d_a, int rows, int cols, double *d_out);
int gpr_zinv(hipStream_t st, const double *d_noise, double sigma, int m, double *d_zinv);
int gpr_rowquad(hipStream_t st, const double *dV2, long ld2, const double *dW, int mnew, int K, const double *d_l, double c,
                double *d_cov);
int gpr_rowdot(hipStream_t st, const double *dC21, const double *dAl, int mnew, int m, const double *dV2, long ld2, int K,
               const double *d_l, double c, double *d_cov);

// host wait for a stream that polls an event instead of sleeping in hipStreamSynchronize (eig.hip)
hipError_t stream_wait(hipStream_t st);

// Register-resident LAE kernels (lae_reg*.hip).  Returns FLGP_LAE_REG_NONE when no kernel of the family
// is built for (r, d) and the caller falls through to the LDS kernels of lae.hip.
#define FLGP_LAE_REG_NONE 1
int launch_lae_reg(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad, int r,
                   const int *d_knn, int ldk, int *d_ei, double *d_ev);

}  // namespace flgp
