// anchors on the device.  The reference's subsample_cpp (src/Utils.cpp:32-68) calls back into R
// (stats::kmeans -- Hartigan-Wong with R's RNG -- or ClusterR::MiniBatchKmeans): neither can be reproduced outside R,
// and at n = 1e6, s = 5000 that call dominates the R wall clock.  This is the plain Lloyd iteration on the path's own
// kernels, offered as an extra `subsample` method, not as a bit-for-bit stand-in for R's k-means:
//   assign  : 1-NN of every point to the current centres -- the k-NN kernel with r = 1 (same arithmetic, ties to
//             the lower centre index), which is also how the reference counts cluster sizes (src/Utils.cpp:59-62);
//   update  : centre = (sum of its points, added in row order) / count -- the CSC build gives every centre its rows
//             ascending, so the sums are deterministic; a centre that lost all its points keeps its position, size 0;
//   stop    : when no label changed, or after iter_max assign/update rounds.
// The CPU restatement with the same operation order is oracle.np_kmeans_lloyd; results are compared bit for bit.
#include "common.h"

namespace flgp {

// one thread per (centre c, coordinate k): sequential sum over the rows of c in ascending order, then IEEE division
__global__ __launch_bounds__(64) void km_update_kernel(const double *__restrict__ X, int ldx, const int *__restrict__ colptr,
                                                       const int *__restrict__ rows, int s, double *__restrict__ C, int ldc,
                                                       double *__restrict__ size) {
  const int c = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (c >= s) return;
  const int p0 = colptr[c], p1 = colptr[c + 1];
  const double *xk = X + (size_t)k * ldx;
  double a = 0.0;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    const double v0 = xk[rows[p]], v1 = xk[rows[p + 1]], v2 = xk[rows[p + 2]], v3 = xk[rows[p + 3]];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; p < p1; ++p) a += xk[rows[p]];
  if (p1 > p0) C[(size_t)k * ldc + c] = a / (double)(p1 - p0);
  if (k == 0) size[c] = (double)(p1 - p0);
}

// changed += #{i : lab[i] != prev[i]};  prev <- lab
__global__ void km_changed_kernel(const int *__restrict__ lab, int *__restrict__ prev, int n, int *__restrict__ changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool diff = i < n && lab[i] != prev[i];
  if (i < n) prev[i] = lab[i];
  const unsigned long long m = __ballot(diff);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(changed, __popcll(m));
}

__global__ void km_gather_rows_kernel(const double *__restrict__ X, int ldx, int d, const int *__restrict__ rows, int s,
                                      double *__restrict__ C, int ldc) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * d) return;
  const int c = (int)(e % s), k = (int)(e / s);
  C[(size_t)k * ldc + c] = X[(size_t)k * ldx + rows[c]];
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_anchor_dpad(int d);
extern "C" int flgp_dev_anchor_rows(int s);
extern "C" int flgp_dev_anchor_prep(void *stream, const double *dU, int s, int ldu, int d, double *dUt, double *duu);
extern "C" int flgp_dev_knn(void *stream, const double *dX, int n, int ldx, int d, const double *dUt, const double *duu,
                            int s, int r, int *d_idx, double *d_dist, int ldo);
extern "C" size_t flgp_dev_csc_workspace(int n, int s, int r);
extern "C" int flgp_dev_csc_build(void *stream, const int *d_ell_idx, int n, int s, int r, int *d_colptr, int *d_pos,
                                  void *d_work, size_t work_bytes);
extern "C" int flgp_dev_mean(void *stream, const double *d_x, long count, double *d_out, double *d_work);

// dX: n x d column-major (ldx).  dC: s x d column-major (ldc): initial centres in, final centres out.  d_size: s.
// iters_out: assign/update rounds performed; withinss_out (optional): sum over points of the squared distance to the
// centre they were last assigned to.  Synchronous (one 4-byte readback per round).
extern "C" int flgp_dev_kmeans_lloyd(void *stream, const double *dX, int n, int ldx, int d, int s, double *dC, int ldc,
                                     double *d_size, int iter_max, int *iters_out, double *withinss_out) {
  hipStream_t st = (hipStream_t)stream;
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "kmeans: kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(n >= 1 && s >= 1 && s <= n && iter_max >= 1, "kmeans: need 1 <= s <= n and iter_max >= 1");
  FLGP_REQUIRE(ldx >= n && ldc >= s, "kmeans: leading dimensions too small");
  const int rows = flgp_dev_anchor_rows(s);
  DevBuf Ut, uu, lab, prev, colptr, pos, work, flag, dist, mwork, mean;
  FLGP_TRY(Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(lab.alloc(sizeof(int) * (size_t)n));
  FLGP_TRY(prev.alloc(sizeof(int) * (size_t)n));
  FLGP_TRY(colptr.alloc(sizeof(int) * (size_t)(s + 1)));
  FLGP_TRY(pos.alloc(sizeof(int) * (size_t)n));
  const size_t wb = flgp_dev_csc_workspace(n, s, 1);
  FLGP_TRY(work.alloc(wb));
  FLGP_TRY(flag.alloc(sizeof(int)));
  if (withinss_out) {
    FLGP_TRY(dist.alloc(sizeof(double) * (size_t)n));
    FLGP_TRY(mwork.alloc(sizeof(double) * (size_t)(ceil_div((long)n, 4096) + 1)));
    FLGP_TRY(mean.alloc(sizeof(double)));
  }
  FLGP_HIP(hipMemsetAsync(prev.p, 0xff, sizeof(int) * (size_t)n, st));   // label -1: everything "changes" in round 1
  int it = 0;
  for (;;) {
    FLGP_TRY(flgp_dev_anchor_prep(st, dC, s, ldc, d, Ut.as<double>(), uu.as<double>()));
    FLGP_TRY(flgp_dev_knn(st, dX, n, ldx, d, Ut.as<double>(), uu.as<double>(), s, 1, lab.as<int>(),
                          withinss_out ? dist.as<double>() : nullptr, n));
    FLGP_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(km_changed_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, lab.as<int>(), prev.as<int>(), n, flag.as<int>());
    FLGP_TRY(check_launch("km_changed_kernel"));
    int changed = 0;
    FLGP_HIP(hipMemcpyAsync(&changed, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    if (changed == 0) break;          // the centres already are the means of this labelling
    FLGP_TRY(flgp_dev_csc_build(st, lab.as<int>(), n, s, 1, colptr.as<int>(), pos.as<int>(), work.p, wb));
    hipLaunchKernelGGL(km_update_kernel, dim3(ceil_div(s, 64), d), dim3(64), 0, st, dX, ldx, colptr.as<int>(), pos.as<int>(), s,
                       dC, ldc, d_size);
    FLGP_TRY(check_launch("km_update_kernel"));
    if (++it >= iter_max) break;
  }
  if (withinss_out) {
    FLGP_TRY(flgp_dev_mean(st, dist.as<double>(), n, mean.as<double>(), mwork.as<double>()));
    double m = 0.0;
    FLGP_HIP(hipMemcpyAsync(&m, mean.p, sizeof(double), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    *withinss_out = m * (double)n;
  }
  FLGP_HIP(hipStreamSynchronize(st));
  if (iters_out) *iters_out = it;
  return FLGP_OK;
}

extern "C" int flgp_dev_kmeans_init(void *stream, const double *dX, int n, int ldx, int d, const int *d_rows, int s,
                                    double *dC, int ldc) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(n >= 1 && d >= 1 && s >= 1 && ldx >= n && ldc >= s, "kmeans_init: bad shape");
  hipLaunchKernelGGL(km_gather_rows_kernel, dim3(ceil_div((long)s * d, 256)), dim3(256), 0, st, dX, ldx, d, d_rows, s, dC, ldc);
  return check_launch("km_gather_rows_kernel");
}
