// k-NN for d > 64 (KNN_cpp / KNN_Index, reference src/Utils.cpp:72-192, which have no limit on the dimension).
//
// The kernels of knn.hip keep a point's coordinates in registers, which ends at d = 64.  Beyond that the distance block
// is what the reference says it is -- a GEMM, src/Utils.cpp:121 -- and is computed as one, row block by row block:
//   dots = X_b U^T          gemm.hip without split-K: every element one k-ascending FMA chain from C = 0 on the matrix
//                           cores, i.e. the oracle's  dot = x0 u0; dot = fma(x_k, u_k, dot)  bit for bit (knn.hip, the
//                           note on v_mfma_f64_16x16x4_f64; the operand swap inside gemm_launch only commutes products)
//   D = fma(-2, dot, |x|^2) + |u|^2, one lane per point scanning the anchors in ascending order with the strict '<'
//                           of TopList: the lower index keeps a tie, as in the oracle's scan.
// The row block is sized so that the n_b x s block of dot products stays under 1 GB.
#include "common.h"
#include "knn_top.h"

namespace flgp {

// |x|^2 by the oracle's chain (first term a product, then FMAs in ascending k)
__global__ void knn_wide_sqnorm_kernel(const double *__restrict__ X, int nb, int ldx, int d, double *__restrict__ xx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nb) return;
  double a = X[x] * X[x];
  for (int k = 1; k < d; ++k) {
    const double v = X[(size_t)k * ldx + x];
    a = __builtin_fma(v, v, a);
  }
  xx[x] = a;
}

template <int RCAP>
__global__ __launch_bounds__(256) void knn_wide_select_kernel(const double *__restrict__ dots, int nb,
                                                              const double *__restrict__ xx,
                                                              const double *__restrict__ uu, int s, int r,
                                                              int *__restrict__ idx_out, double *__restrict__ dist_out,
                                                              int ldo) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int xc = x < nb ? x : nb - 1;
  const double xn = xx[xc];
  TopList<RCAP> top;
  top.init(r);
  const double *col = dots + xc;
  // four anchors' loads in flight per step; the order of the comparisons stays ascending
  int j = 0;
  for (; j + 4 <= s; j += 4) {
    double dv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dv[q] = col[(size_t)(j + q) * nb];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double D = __builtin_fma(-2.0, dv[q], xn) + uu[j + q];
      if (D < top.thr()) top.insert(D, j + q);
    }
  }
  for (; j < s; ++j) {
    const double D = __builtin_fma(-2.0, col[(size_t)j * nb], xn) + uu[j];
    if (D < top.thr()) top.insert(D, j);
  }
  if (x < nb) {
#pragma unroll
    for (int k = 0; k < RCAP; ++k) {
      const int slot = k - (RCAP - r);
      if (slot >= 0) {
        const int bj_ = top.bi[k];     // as in knn_kernel: never let the sentinel out (rows of NaN select nothing)
        idx_out[(size_t)slot * ldo + x] = ((unsigned)bj_ < (unsigned)s) ? bj_ : slot;
        if (dist_out) dist_out[(size_t)slot * ldo + x] = top.bd[k];
      }
    }
  }
}

// rows of X per block of dot products
static int knn_wide_block(int n, int s) {
  long nb = ((long)1 << 27) / s / 256 * 256;
  const long forced = tuning("knn_wide_block", 0);   // tests: several blocks on small inputs
  if (forced > 0) nb = forced;
  if (nb < 256) nb = 256;
  if (nb > n) nb = n;
  return (int)nb;
}

int knn_wide(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad, const double *duu, int s,
             int r, int *d_idx, double *d_dist, int ldo) {
  const int NB = knn_wide_block(n, s);
  DevBuf dots, xx;
  FLGP_TRY(dots.alloc(sizeof(double) * (size_t)NB * s));
  FLGP_TRY(xx.alloc(sizeof(double) * (size_t)NB));
  const int rcap = r <= 4 ? 4 : (r <= 8 ? 8 : (r <= 16 ? 16 : 32));
  for (int x0 = 0; x0 < n; x0 += NB) {
    const int nb = (n - x0 < NB) ? n - x0 : NB;
    hipLaunchKernelGGL(knn_wide_sqnorm_kernel, dim3(ceil_div(nb, 256)), dim3(256), 0, st, dX + x0, nb, ldx, d, xx.as<double>());
    FLGP_TRY(check_launch("knn_wide_sqnorm_kernel"));
    // dots(x, j) = sum_k X(x0 + x, k) Ut(j, k): no workspace, so no split-K -- one chain per element
    FLGP_TRY(gemm_launch(st, nb, s, d, 1.0, dX + x0, 1, ldx, dUt, 1, dpad, 0.0, nullptr, 0, 0, dots.as<double>(), 1, nb,
                         nullptr, 0, 0.0, nullptr));
    {
      ProfScope ps("knn_wide_select", st, 3.0 * (double)nb * (double)s);
      const dim3 grid(ceil_div(nb, 256));
      int *io = d_idx + x0;
      double *dd = d_dist ? d_dist + x0 : nullptr;
#define KNN_WIDE_CASE(RCv)                                                                                             \
  if (rcap == RCv)                                                                                                     \
    hipLaunchKernelGGL((knn_wide_select_kernel<RCv>), grid, dim3(256), 0, st, dots.as<double>(), nb, xx.as<double>(), \
                       duu, s, r, io, dd, ldo);
      KNN_WIDE_CASE(4) KNN_WIDE_CASE(8) KNN_WIDE_CASE(16) KNN_WIDE_CASE(32)
#undef KNN_WIDE_CASE
    }
    FLGP_TRY(check_launch("knn_wide_select_kernel"));
  }
  return FLGP_OK;   // (the blocks go back to the cache: that synchronises the device)
}

}  // namespace flgp
