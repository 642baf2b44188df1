// k1 + k2: fused pairwise squared distance + top-r selection (k-NN of n points to s anchors).
//
// Replaces KNN_cpp / KNN_Index, reference src/Utils.cpp:72-192:
//   D = ((-2 X_b U^T).colwise() + |x|^2).rowwise() + |u|^2      (:121)
//   per row: std::partial_sort of an index array by D, first r   (:91-94)
// The reference walks 100-row GEMM batches and materialises the n x s distance block; here
// one lane owns P points (their coordinates live in VGPRs for the whole kernel), anchors
// stream past as wave-uniform operands -- either broadcast reads of an LDS-staged anchor
// tile or scalar (SGPR) loads -- and the distance block is never written anywhere.
//
// Arithmetic (identical, operation for operation, to oracle/flgp_oracle.c so that indices AND
// distances agree bit for bit): dot = x0*u0; dot = fma(x_k, u_k, dot) k = 1..d-1;
// D = fma(-2, dot, |x|^2) + |u|^2.  Ties: lower anchor index wins.
//
// Selection: each lane keeps a sorted top-r list in registers.  A candidate is first tested
// against the lane's current r-th best (one v_cmp_lt_f64); survivors are appended to a small
// per-lane LDS queue, and the whole wave drains its queues into the sorted lists only when
// some lane's queue is full (wave-wide ballot).  That keeps the ~r ln(s/r) insertions per
// point from turning into a divergent branch on every candidate: a wave with 64 lanes would
// otherwise take the insertion path for a third of all candidates at s = 5000, r = 10.
#include "common.h"

namespace flgp {

__global__ void anchor_prep_kernel(const double *__restrict__ U, int s, int s_pad, int ldu, int d, int dpad,
                                   double *__restrict__ Ut, double *__restrict__ uu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s_pad) return;
  if (j >= s) {  // padding rows: never selected (|u|^2 = +inf makes D = +inf, and inf < thr is false)
    for (int k = 0; k < dpad; ++k) Ut[(size_t)j * dpad + k] = 0.0;
    uu[j] = __builtin_inf();
    return;
  }
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    const double u = U[(size_t)k * ldu + j];
    Ut[(size_t)j * dpad + k] = u;
    acc = (k == 0) ? u * u : __builtin_fma(u, u, acc);
  }
  for (int k = d; k < dpad; ++k) Ut[(size_t)j * dpad + k] = 0.0;
  uu[j] = acc;
}

// Sorted list of the RCAP smallest (value, index) pairs, ascending.  The first RCAP - r slots
// are pinned by -inf sentinels so that the r-th best real candidate is always bd[RCAP-1]
// (a compile-time register), whatever the run-time r.
template <int RCAP>
struct TopList {
  double bd[RCAP];
  int bi[RCAP];
  __device__ __forceinline__ void init(int r) {
#pragma unroll
    for (int k = 0; k < RCAP; ++k) {
      bd[k] = (k < RCAP - r) ? -__builtin_inf() : __builtin_inf();
      bi[k] = (k < RCAP - r) ? -1 : 0x7fffffff;
    }
  }
  __device__ __forceinline__ double thr() const { return bd[RCAP - 1]; }
  // strict '<': an equal distance never moves ahead of an earlier (lower) index
  __device__ __forceinline__ void insert(double D, int j) {
#pragma unroll
    for (int k = RCAP - 1; k >= 1; --k) {
      const bool c1 = D < bd[k - 1];
      const bool c0 = D < bd[k];
      bd[k] = c1 ? bd[k - 1] : (c0 ? D : bd[k]);
      bi[k] = c1 ? bi[k - 1] : (c0 ? j : bi[k]);
    }
    const bool c0 = D < bd[0];
    bd[0] = c0 ? D : bd[0];
    bi[0] = c0 ? j : bi[0];
  }
};

template <int DP>
constexpr int knn_tile_anchors() { return DP <= 16 ? 128 : (DP <= 32 ? 64 : 32); }

// P points per lane x A anchors per step = P*A independent FMA chains per wave (a dependent
// v_fma_f64 chain alone reaches ~1/7 of the fp64 VALU rate on gfx950: scripts/ubench_fma64.hip).
// The first KS coordinates of every anchor come in as SGPR operands (wave-uniform scalar loads
// straight from the padded anchor panel), the remaining DP-KS from broadcast reads of the LDS tile:
// splitting the operand stream keeps both the LDS pipe and the scalar cache under their limits.
template <int DP, int RCAP, int P, int A, int KS, int QC>
__global__ __launch_bounds__(256) void knn_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                  const double *__restrict__ Ut,
                                                  const double *__restrict__ uu, int s, int r,
                                                  int *__restrict__ idx_out,
                                                  double *__restrict__ dist_out, int ldo) {
  constexpr int NT = 256;
  constexpr int TA = knn_tile_anchors<DP>();
  constexpr bool USE_LDS = KS < DP;
  static_assert(TA % A == 0 && QC > A, "tile / queue geometry");
  __shared__ double q_d[QC * P * NT];
  __shared__ int q_j[QC * P * NT];
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? TA * DP : 2];

  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * (NT * P);

  double x[P][DP];
  double xx[P];
  TopList<RCAP> top[P];
  int cnt[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    long i = base + (long)p * NT + tid;
    if (i >= n) i = n - 1;  // clamp: computes a duplicate, never stored
#pragma unroll
    for (int k = 0; k < DP; ++k) x[p][k] = (k < d) ? X[(size_t)k * ldx + i] : 0.0;
    double acc = x[p][0] * x[p][0];
#pragma unroll
    for (int k = 1; k < DP; ++k) acc = __builtin_fma(x[p][k], x[p][k], acc);  // zero padding adds exactly 0
    xx[p] = acc;
    top[p].init(r);
    cnt[p] = 0;
  }

  auto drain = [&]() {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int c = cnt[p];
      for (int q = 0; __any(q < c); ++q) {
        if (q < c) {
          const double D = q_d[(q * P + p) * NT + tid];
          const int j = q_j[(q * P + p) * NT + tid];
          if (D < top[p].thr()) top[p].insert(D, j);
        }
      }
      cnt[p] = 0;
    }
  };

  // the anchor panel is padded to a multiple of TA rows (zeros, |u|^2 = +inf), so every
  // step may touch A full rows
  const int s_pad = (s + TA - 1) / TA * TA;
  for (int j0 = 0; j0 < s_pad; j0 += TA) {
    if (USE_LDS) {
      __syncthreads();
      const double *src = Ut + (size_t)j0 * DP;
      for (int e = tid; e < TA * DP; e += NT) tile[e] = src[e];
      __syncthreads();
    }
    for (int jj = 0; jj < TA; jj += A) {
      const double *__restrict__ ug = Ut + (size_t)(j0 + jj) * DP;  // wave-uniform -> s_load
      const double *ul = tile + jj * DP;                              // wave-uniform -> broadcast ds_read
      // the A squared norms in one scalar load at the top of the step: fetched one by one between
      // the compare/branch blocks they each cost an exposed scalar-cache round trip
      double uug[A];
#pragma unroll
      for (int a = 0; a < A; ++a) uug[a] = uu[j0 + jj + a];
      double acc[P][A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const double u0 = (0 < KS) ? ug[a * DP] : ul[a * DP];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][a] = x[p][0] * u0;
      }
#pragma unroll
      for (int k = 1; k < DP; ++k) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
          const double uk = (k < KS) ? ug[a * DP + k] : ul[a * DP + k];
#pragma unroll
          for (int p = 0; p < P; ++p) acc[p][a] = __builtin_fma(x[p][k], uk, acc[p][a]);
        }
      }
      bool full = false;
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const double uuj = uug[a];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const double D = __builtin_fma(-2.0, acc[p][a], xx[p]) + uuj;
          if (D < top[p].thr()) {
            q_d[(cnt[p] * P + p) * NT + tid] = D;
            q_j[(cnt[p] * P + p) * NT + tid] = j0 + jj + a;
            ++cnt[p];
          }
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) full |= (cnt[p] > QC - A);
      if (__any(full)) drain();
    }
  }
  drain();

#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long i = base + (long)p * NT + tid;
    if (i < n) {
#pragma unroll
      for (int k = 0; k < RCAP; ++k) {
        const int slot = k - (RCAP - r);
        if (slot >= 0) {
          idx_out[(size_t)slot * ldo + i] = top[p].bi[k];
          if (dist_out) dist_out[(size_t)slot * ldo + i] = top[p].bd[k];
        }
      }
    }
  }
}

template <int DP, int RCAP, int P, int A, int KS, int QC>
static int launch_knn(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt,
                      const double *duu, int s, int r, int *d_idx, double *d_dist, int ldo) {
  const int grid = ceil_div(n, 256 * P);
  ProfScope ps("knn_kernel", st, 2.0 * (double)n * (double)s * (double)d);
  hipLaunchKernelGGL((knn_kernel<DP, RCAP, P, A, KS, QC>), dim3(grid), dim3(256), 0, st, dX, n, ldx, d,
                     dUt, duu, s, r, d_idx, d_dist, ldo);
  return check_launch("knn_kernel");
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_anchor_dpad(int d) {
  if (d <= 4) return 4;
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  return -1;
}

// rows of the padded anchor panel: s rounded up to the kernel's anchor tile
extern "C" int flgp_dev_anchor_rows(int s) { return (s + 127) / 128 * 128; }

extern "C" int flgp_dev_anchor_prep(void *stream, const double *dU, int s, int ldu, int d, double *dUt,
                                    double *duu) {
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "k-NN kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(s >= 1 && ldu >= s, "anchor_prep: need s >= 1 and ldu >= s");
  const int s_pad = flgp_dev_anchor_rows(s);
  hipLaunchKernelGGL(anchor_prep_kernel, dim3(ceil_div(s_pad, 256)), dim3(256), 0, (hipStream_t)stream, dU, s,
                     s_pad, ldu, d, dpad, dUt, duu);
  return check_launch("anchor_prep_kernel");
}

#define KNN_ARGS st, dX, n, ldx, d, dUt, duu, s, r, d_idx, d_dist, ldo
// default shape per padded dimension: one point per lane, A anchors per step, the first KS
// coordinates as SGPR operands (measured on MI355X at n=1e6, d=16, s=5000, r=10:
// P1/A2/KS8 5.8 ms, P1/A4/KS8 6.7 ms, P1/A4/KS0 7.4 ms, P1/A2/KS16 10.7 ms, P2/A4/KS8 10.6 ms: fewer
// VGPRs (3 waves/SIMD) beat more chains per wave; an all-scalar operand stream starves on the scalar cache)
#define KNN_CASE(DPv, RCv)                                                                   \
  if (dpad == DPv && rcap == RCv) {                                                          \
    if constexpr (DPv == 16) return launch_knn<DPv, RCv, 1, 2, 8, 8>(KNN_ARGS);              \
    else if constexpr (DPv < 16) return launch_knn<DPv, RCv, 1, 4, DPv / 2, 12>(KNN_ARGS);   \
    else return launch_knn<DPv, RCv, 1, 2, 16, 8>(KNN_ARGS);                                 \
  }

extern "C" int flgp_dev_knn(void *stream, const double *dX, int n, int ldx, int d, const double *dUt,
                            const double *duu, int s, int r, int *d_idx, double *d_dist, int ldo) {
  hipStream_t st = (hipStream_t)stream;
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "k-NN kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(r >= 1 && r <= s, "KNN: need 1 <= r <= s (r=%d, s=%d)", r, s);
  FLGP_REQUIRE(r <= FLGP_RMAX, "KNN: r=%d exceeds the built maximum %d", r, FLGP_RMAX);
  FLGP_REQUIRE(ldx >= n && ldo >= n, "KNN: leading dimensions must be >= n");
  if (n == 0) return FLGP_OK;
  const int rcap = r <= 4 ? 4 : (r <= 8 ? 8 : (r <= 16 ? 16 : 32));
  const int variant = tuning("knn_variant", 0);
  // experimental shapes, d <= 16 and r <= 16 only (selected through flgp_set_tuning)
  if (dpad == 16 && rcap == 16) {
    if (variant == 1) return launch_knn<16, 16, 2, 4, 8, 12>(KNN_ARGS);
    if (variant == 2) return launch_knn<16, 16, 2, 2, 16, 8>(KNN_ARGS);
    if (variant == 4) return launch_knn<16, 16, 1, 4, 0, 12>(KNN_ARGS);
    if (variant == 5) return launch_knn<16, 16, 1, 2, 16, 8>(KNN_ARGS);
    if (variant == 6) return launch_knn<16, 16, 1, 2, 8, 8>(KNN_ARGS);
    if (variant == 7) return launch_knn<16, 16, 1, 2, 4, 8>(KNN_ARGS);
    if (variant == 8) return launch_knn<16, 16, 1, 2, 12, 8>(KNN_ARGS);
    if (variant == 9) return launch_knn<16, 16, 1, 1, 8, 8>(KNN_ARGS);
    if (variant == 10) return launch_knn<16, 16, 1, 2, 0, 8>(KNN_ARGS);
  }
  KNN_CASE(4, 4) KNN_CASE(4, 8) KNN_CASE(4, 16) KNN_CASE(4, 32)
  KNN_CASE(8, 4) KNN_CASE(8, 8) KNN_CASE(8, 16) KNN_CASE(8, 32)
  KNN_CASE(16, 4) KNN_CASE(16, 8) KNN_CASE(16, 16) KNN_CASE(16, 32)
  KNN_CASE(32, 4) KNN_CASE(32, 8) KNN_CASE(32, 16) KNN_CASE(32, 32)
  KNN_CASE(64, 4) KNN_CASE(64, 8) KNN_CASE(64, 16) KNN_CASE(64, 32)
  set_error("KNN: no kernel for dpad=%d rcap=%d", dpad, rcap);
  return FLGP_ERR_INVALID;
}
