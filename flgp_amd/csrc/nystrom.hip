// the Nystrom-extension spectrum of the fit_nystrom_* drivers (reference src/Fit.cpp:244-289, the same
// block in :399-441, :918-960, :1063-1105, :1222-1264, :1379-1421), per bandwidth a2:
//   D_UU, D_XU squared distances;  mean = sum(D_UU)/s^2;  Z = exp(-D/(a2 mean))
//   rs_U = rowsum(Z_UU) + 1e-9;  A_UU = Z_UU / (rs_U rs_U^T);  sd = 1/sqrt(rowsum(A_UU) + 1e-9);  W_UU = sd A_UU sd
//   (values, V) = top-K eigenpairs of W_UU;  V <- sd V, columns rescaled to norm sqrt(s) (+1e-9 guards)
//   rs_X = rowsum(Z_XU) + 1e-9;  A_XU = Z_XU / (rs_X rs_U^T);  W_XU = A_XU / (rowsum(A_XU) + 1e-9)
//   vectors = W_XU V diag(1/(|values| + 1e-9))
// The n x s similarity (400 GB at C5) is never materialised: row blocks of Z_XU are produced together with their row
// sums, multiplied with the (pre-scaled) anchor eigenvectors by the MFMA GEMM, and rescaled.  W_UU is dense
// (every anchor sees every other one), so its eigensolve takes the dense-product path of eig.hip.
// Distances use the k-NN arithmetic (k-ascending FMA chain); the reference's come out of an Eigen GEMM whose
// summation order is unspecified, so this stage is compared at rounding level, not bit for bit.
#include "common.h"

namespace flgp {

// MODE 0: out(x, j) = D(x, u_j), acc1 = sum_j D.   MODE 1: out = exp(-D inv_c), acc1 = sum_j out, acc2 = sum_j out w_j.
// Thread x owns one point (coordinates in registers); blockIdx.y owns a chunk of 64 anchors (wave-uniform operands:
// scalar loads of the padded panel).  Partial sums go to part1/part2[chunk][x] and are added in chunk order later.
template <int DP, int MODE>
__global__ __launch_bounds__(256) void nys_sim_kernel(const double *__restrict__ X, int nb, int ldx, int d,
                                                      const double *__restrict__ Ut, const double *__restrict__ uu, int s,
                                                      double inv_c, const double *__restrict__ w, double *__restrict__ out,
                                                      int ldo, double *__restrict__ part1, double *__restrict__ part2) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int xc = x < nb ? x : nb - 1;
  double xv[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) xv[k] = (k < d) ? X[(size_t)k * ldx + xc] : 0.0;
  double xx = xv[0] * xv[0];
#pragma unroll
  for (int k = 1; k < DP; ++k) xx = __builtin_fma(xv[k], xv[k], xx);
  const int j0 = blockIdx.y * 64, j1 = (j0 + 64 < s) ? j0 + 64 : s;
  double a1 = 0.0, a2 = 0.0;
  for (int j = j0; j < j1; ++j) {
    const double *u = Ut + (size_t)j * DP;
    double dot = xv[0] * u[0];
#pragma unroll
    for (int k = 1; k < DP; ++k) dot = __builtin_fma(xv[k], u[k], dot);
    const double D = __builtin_fma(-2.0, dot, xx) + uu[j];
    double v = D;
    if (MODE == 1) v = exp(-D * inv_c);
    if (x < nb) out[(size_t)j * ldo + x] = v;
    a1 += v;
    if (MODE == 1) a2 += v * w[j];
  }
  if (x < nb) {
    part1[(size_t)blockIdx.y * nb + x] = a1;
    if (MODE == 1) part2[(size_t)blockIdx.y * nb + x] = a2;
  }
}

// d > 32: the dot products come from the MFMA GEMM (a k-ascending FMA chain from 0, the same bits as the chain
// above); the 64 anchor coordinates of the scalar-operand kernel no longer fit the SGPR file and its loads stall.
__global__ void nys_sqnorm_kernel(const double *__restrict__ X, int nb, int ldx, int d, double *__restrict__ xx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nb) return;
  double a = X[x] * X[x];
  for (int k = 1; k < d; ++k) a = __builtin_fma(X[(size_t)k * ldx + x], X[(size_t)k * ldx + x], a);
  xx[x] = a;
}
// in place on the dot products: Z(x, j) = exp(-(fma(-2, dot, xx) + uu_j) inv_c), with the two partial row sums
__global__ __launch_bounds__(256) void nys_exp_rows_kernel(double *__restrict__ Z, int nb, int s, const double *__restrict__ xx,
                                                           const double *__restrict__ uu, double inv_c,
                                                           const double *__restrict__ w, double *__restrict__ part1,
                                                           double *__restrict__ part2) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= nb) return;
  const double xn = xx[x];
  const int j0 = blockIdx.y * 64, j1 = (j0 + 64 < s) ? j0 + 64 : s;
  double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
  for (int j = j0; j < j1; ++j) {
    const double D = __builtin_fma(-2.0, Z[(size_t)j * nb + x], xn) + uu[j];
    const double v = exp(-D * inv_c);
    Z[(size_t)j * nb + x] = v;
    a1 += v;
    a2 += v * w[j];
  }
  part1[(size_t)blockIdx.y * nb + x] = a1;
  part2[(size_t)blockIdx.y * nb + x] = a2;
}

// the same on the dot products of the anchors with themselves: Z(x, j) = D(x, j), row sums in part1 (MODE 0 of nys_sim_kernel)
__global__ __launch_bounds__(256) void nys_dist_rows_kernel(double *__restrict__ Z, int nb, int s, const double *__restrict__ xx,
                                                            const double *__restrict__ uu, double *__restrict__ part1) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= nb) return;
  const double xn = xx[x];
  const int j0 = blockIdx.y * 64, j1 = (j0 + 64 < s) ? j0 + 64 : s;
  double a1 = 0.0;
  for (int j = j0; j < j1; ++j) {
    const double D = __builtin_fma(-2.0, Z[(size_t)j * nb + x], xn) + uu[j];
    Z[(size_t)j * nb + x] = D;
    a1 += D;
  }
  part1[(size_t)blockIdx.y * nb + x] = a1;
}

// r1[x] = sum over chunks (ascending) of part1[chunk][x] (+ add1); same for r2
__global__ void nys_reduce_kernel(const double *__restrict__ part1, const double *__restrict__ part2, int nchunk, int nb,
                                  double add1, double *__restrict__ r1, double *__restrict__ r2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nb) return;
  double a = 0.0, b = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    a += part1[(size_t)c * nb + x];
    if (part2) b += part2[(size_t)c * nb + x];
  }
  r1[x] = a + add1;
  if (r2) r2[x] = b;
}

__global__ void nys_exp_kernel(double *__restrict__ M, long count, double inv_c) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) M[e] = exp(-M[e] * inv_c);
}
// colsum[j] = sum_i M(i, j) (rows ascending, one workgroup per column, fixed tree)
__global__ __launch_bounds__(256) void nys_colsum_kernel(const double *__restrict__ M, int s, double *__restrict__ colsum) {
  __shared__ double red[256];
  const int j = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < s; i += 256) a += M[(size_t)j * s + i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) colsum[j] = red[0];
}
// mode 0: v[j] = 1 / (v[j] + 1e-9);  mode 1: v[j] = 1 / sqrt(v[j] + 1e-9)
__global__ void nys_vec_kernel(double *__restrict__ v, int s, int mode) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s) return;
  v[j] = mode == 0 ? 1.0 / (v[j] + 1e-9) : 1.0 / __builtin_sqrt(v[j] + 1e-9);
}
// M(i, j) <- (M(i, j) a[i]) a[j]
__global__ void nys_symscale_kernel(double *__restrict__ M, int s, const double *__restrict__ a) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * s) return;
  const int i = (int)(e % s), j = (int)(e / s);
  M[e] = (M[e] * a[i]) * a[j];
}
// anchor eigenvectors: V(j,k) <- sd[j] V(j,k); norms per column; then the column and row factors of the extension
__global__ void nys_rowscale_kernel(double *__restrict__ V, int rows, int cols, int ld, const double *__restrict__ f) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  const int i = (int)(e % rows), k = (int)(e / rows);
  V[(size_t)k * ld + i] *= f[i];
}
__global__ __launch_bounds__(256) void nys_colnorm_kernel(const double *__restrict__ V, int s, double *__restrict__ nrm) {
  __shared__ double red[256];
  const int k = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < s; i += 256) { const double v = V[(size_t)k * s + i]; a = __builtin_fma(v, v, a); }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) nrm[k] = __builtin_sqrt(red[0]);
}
// V(j,k) <- V(j,k) * sqrt(s)/(nrm_k + 1e-9) * rsu_inv[j] / (|val_k| + 1e-9)
__global__ void nys_vfinal_kernel(double *__restrict__ V, int s, int K, const double *__restrict__ nrm,
                                  const double *__restrict__ rsu_inv, const double *__restrict__ val, double sqrt_s) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * K) return;
  const int j = (int)(e % s), k = (int)(e / s);
  V[e] = ((V[e] * (sqrt_s / (nrm[k] + 1e-9))) * rsu_inv[j]) / (__builtin_fabs(val[k]) + 1e-9);
}
// f[x] = (1/rsx[x]) / (s1[x]/rsx[x] + 1e-9)
__global__ void nys_factor_kernel(const double *__restrict__ rsx, const double *__restrict__ s1, int nb, double *__restrict__ f) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nb) return;
  const double inv = 1.0 / rsx[x];
  f[x] = inv / (s1[x] * inv + 1e-9);
}

template <int MODE>
static int launch_sim(hipStream_t st, int dpad, const double *X, int nb, int ldx, int d, const double *Ut, const double *uu,
                      int s, double inv_c, const double *w, double *out, int ldo, double *p1, double *p2) {
  const dim3 grid(ceil_div(nb, 256), ceil_div(s, 64));
#define NYS_CASE(DPv) if (dpad == DPv) hipLaunchKernelGGL((nys_sim_kernel<DPv, MODE>), grid, dim3(256), 0, st, X, nb, ldx, d, Ut, uu, s, inv_c, w, out, ldo, p1, p2);
  NYS_CASE(4) NYS_CASE(8) NYS_CASE(16) NYS_CASE(32) NYS_CASE(64)
#undef NYS_CASE
  FLGP_REQUIRE(dpad <= 64, "nystrom: no register kernel for dpad = %d (the GEMM route serves d > 64)", dpad);
  return check_launch("nys_sim_kernel");
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_anchor_dpad(int d);
extern "C" int flgp_dev_anchor_rows(int s);
extern "C" int flgp_dev_anchor_prep(void *stream, const double *dU, int s, int ldu, int d, double *dUt, double *duu);
extern "C" size_t flgp_dev_eig_workspace(int s, int K);
extern "C" int flgp_dev_eig_topk(void *stream, const double *dG, int ldg, int s, int K, double tol, double *d_values,
                                 double *dV, int ldv, void *d_work, size_t work_bytes, int *info);

// dX: n x d column-major (ldx), dU: s x d column-major (ldu); d_values: K, d_vectors: n x K column-major (ldv).
extern "C" int flgp_dev_nystrom_eigenpair(void *stream, const double *dX, int n, int ldx, int d, const double *dU, int s,
                                          int ldu, double a2, int K, double *d_values, double *d_vectors, int ldv) {
  hipStream_t st = (hipStream_t)stream;
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "nystrom: kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(n >= 1 && s >= 2 && K >= 1 && K <= s && a2 > 0.0, "nystrom: need n >= 1, 1 <= K <= s, a2 > 0");
  FLGP_REQUIRE(ldx >= n && ldu >= s && ldv >= n, "nystrom: leading dimensions too small");
  const int rows = flgp_dev_anchor_rows(s);
  const int nchunk = ceil_div(s, 64);
  // rows of X per block: whole rounds of the GEMM grid (128 x 128 tiles, two co-resident workgroups per CU, 256 CUs)
  // while the block of Z_XU (nb x s) stays under 2 GB
  const long round_rows = 128L * 512 / ceil_div(K, 128);
  const long cap_rows = ((long)256 << 20) / s;
  long nb_max = cap_rows / round_rows * round_rows;
  if (nb_max < round_rows) nb_max = cap_rows / 256 * 256;
  if (nb_max < 256) nb_max = 256;
  if (nb_max > n) nb_max = n;
  const int NB = (int)nb_max;
  const int pmax = NB > s ? NB : s;
  DevBuf Ut, uu, W, Zb, p1, p2, rsu, sd, rsx, s1, fac, eigv, nrm, work, gws;
  FLGP_TRY(Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(W.alloc(sizeof(double) * (size_t)s * s));
  FLGP_TRY(Zb.alloc(sizeof(double) * (size_t)NB * s));
  FLGP_TRY(p1.alloc(sizeof(double) * (size_t)nchunk * pmax));
  FLGP_TRY(p2.alloc(sizeof(double) * (size_t)nchunk * pmax));
  FLGP_TRY(rsu.alloc(sizeof(double) * (size_t)s));
  FLGP_TRY(sd.alloc(sizeof(double) * (size_t)s));
  FLGP_TRY(rsx.alloc(sizeof(double) * (size_t)pmax));
  FLGP_TRY(s1.alloc(sizeof(double) * (size_t)pmax));
  FLGP_TRY(fac.alloc(sizeof(double) * (size_t)pmax));
  FLGP_TRY(eigv.alloc(sizeof(double) * (size_t)s * K));
  FLGP_TRY(nrm.alloc(sizeof(double) * (size_t)K));
  FLGP_TRY(flgp_dev_anchor_prep(st, dU, s, ldu, d, Ut.as<double>(), uu.as<double>()));
  // ---- D_UU and its mean (src/Fit.cpp:244,248)
  if (dpad > 64) {   // no register kernel beyond d = 64: dot products by the GEMM (one chain per element, the same bits)
    hipLaunchKernelGGL(nys_sqnorm_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, dU, s, ldu, d, fac.as<double>());
    FLGP_TRY(check_launch("nys_sqnorm_kernel"));
    FLGP_TRY(gemm_launch(st, s, s, d, 1.0, dU, 1, ldu, dU, ldu, 1, 0.0, nullptr, 0, 0, W.as<double>(), 1, s, nullptr, 0, 0.0,
                         nullptr));
    hipLaunchKernelGGL(nys_dist_rows_kernel, dim3(ceil_div(s, 256), nchunk), dim3(256), 0, st, W.as<double>(), s, s,
                       fac.as<double>(), uu.as<double>(), p1.as<double>());
    FLGP_TRY(check_launch("nys_dist_rows_kernel"));
  } else {
    FLGP_TRY((launch_sim<0>(st, dpad, dU, s, ldu, d, Ut.as<double>(), uu.as<double>(), s, 0.0, nullptr, W.as<double>(), s,
                            p1.as<double>(), nullptr)));
  }
  hipLaunchKernelGGL(nys_reduce_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, p1.as<double>(), nullptr, nchunk, s, 0.0,
                     rsx.as<double>(), nullptr);
  FLGP_TRY(check_launch("nys_reduce_kernel"));
  std::vector<double> hrow(s);
  FLGP_HIP(hipMemcpyAsync(hrow.data(), rsx.p, sizeof(double) * s, hipMemcpyDeviceToHost, st));
  FLGP_HIP(hipStreamSynchronize(st));
  double total = 0.0;
  for (int i = 0; i < s; ++i) total += hrow[i];
  const double mean = total / ((double)s * (double)s);
  FLGP_REQUIRE(mean > 0.0 && std::isfinite(mean), "nystrom: the anchors coincide (mean squared distance %g)", mean);
  const double inv_c = 1.0 / (a2 * mean);
  // ---- Z_UU, rs_U, A_UU, sd, W_UU (:266-270); W is symmetric, so column sums are row sums
  const long ss = (long)s * s;
  hipLaunchKernelGGL(nys_exp_kernel, dim3(ceil_div(ss, 256)), dim3(256), 0, st, W.as<double>(), ss, inv_c);
  hipLaunchKernelGGL(nys_colsum_kernel, dim3(s), dim3(256), 0, st, W.as<double>(), s, rsu.as<double>());
  hipLaunchKernelGGL(nys_vec_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, rsu.as<double>(), s, 0);      // 1/(rs_U + 1e-9)
  hipLaunchKernelGGL(nys_symscale_kernel, dim3(ceil_div(ss, 256)), dim3(256), 0, st, W.as<double>(), s, rsu.as<double>());
  hipLaunchKernelGGL(nys_colsum_kernel, dim3(s), dim3(256), 0, st, W.as<double>(), s, sd.as<double>());
  hipLaunchKernelGGL(nys_vec_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, sd.as<double>(), s, 1);       // 1/sqrt(. + 1e-9)
  hipLaunchKernelGGL(nys_symscale_kernel, dim3(ceil_div(ss, 256)), dim3(256), 0, st, W.as<double>(), s, sd.as<double>());
  FLGP_TRY(check_launch("nystrom W_UU"));
  // ---- top-K eigenpairs of W_UU (eigs_sym, :272-276)
  const size_t wb = flgp_dev_eig_workspace(s, K);
  FLGP_TRY(work.alloc(wb));
  FLGP_TRY(flgp_dev_eig_topk(st, W.as<double>(), s, s, K, 0.0, d_values, eigv.as<double>(), s, work.p, wb, nullptr));
  // ---- V <- sd V, columns to norm sqrt(s) (:278-280), with the two diagonal factors of the extension folded in
  hipLaunchKernelGGL(nys_rowscale_kernel, dim3(ceil_div((long)s * K, 256)), dim3(256), 0, st, eigv.as<double>(), s, K, s,
                     sd.as<double>());
  hipLaunchKernelGGL(nys_colnorm_kernel, dim3(K), dim3(256), 0, st, eigv.as<double>(), s, nrm.as<double>());
  hipLaunchKernelGGL(nys_vfinal_kernel, dim3(ceil_div((long)s * K, 256)), dim3(256), 0, st, eigv.as<double>(), s, K,
                     nrm.as<double>(), rsu.as<double>(), d_values, std::sqrt((double)s));
  FLGP_TRY(check_launch("nystrom V_UU"));
  // ---- extension, row block by row block (:283-289)
  // without a workspace gemm_launch cannot split k, so every dot product is one whole chain
  const bool via_gemm = tuning("nystrom_dot_gemm", -1) == 1 || (tuning("nystrom_dot_gemm", -1) < 0 && dpad > 32);
  const size_t gws_elems = (size_t)8 * NB * K;
  FLGP_TRY(gws.alloc(sizeof(double) * gws_elems));
  for (int x0 = 0; x0 < n; x0 += NB) {
    const int nb = (n - x0 < NB) ? n - x0 : NB;
    if (via_gemm) {
      hipLaunchKernelGGL(nys_sqnorm_kernel, dim3(ceil_div(nb, 256)), dim3(256), 0, st, dX + x0, nb, ldx, d, fac.as<double>());
      FLGP_TRY(check_launch("nys_sqnorm_kernel"));
      FLGP_TRY(gemm_launch(st, nb, s, d, 1.0, dX + x0, 1, ldx, dU, ldu, 1, 0.0, nullptr, 0, 0, Zb.as<double>(), 1, nb,
                           nullptr, 0, 0.0, nullptr));
      hipLaunchKernelGGL(nys_exp_rows_kernel, dim3(ceil_div(nb, 256), nchunk), dim3(256), 0, st, Zb.as<double>(), nb, s,
                         fac.as<double>(), uu.as<double>(), inv_c, rsu.as<double>(), p1.as<double>(), p2.as<double>());
      FLGP_TRY(check_launch("nys_exp_rows_kernel"));
    } else {
      FLGP_TRY((launch_sim<1>(st, dpad, dX + x0, nb, ldx, d, Ut.as<double>(), uu.as<double>(), s, inv_c, rsu.as<double>(),
                              Zb.as<double>(), nb, p1.as<double>(), p2.as<double>())));
    }
    hipLaunchKernelGGL(nys_reduce_kernel, dim3(ceil_div(nb, 256)), dim3(256), 0, st, p1.as<double>(), p2.as<double>(), nchunk,
                       nb, 1e-9, rsx.as<double>(), s1.as<double>());
    hipLaunchKernelGGL(nys_factor_kernel, dim3(ceil_div(nb, 256)), dim3(256), 0, st, rsx.as<double>(), s1.as<double>(), nb,
                       fac.as<double>());
    FLGP_TRY(check_launch("nystrom block sums"));
    // out(x, k) = sum_j Z(x, j) V'(j, k)
    FLGP_TRY(gemm_launch(st, nb, K, s, 1.0, Zb.as<double>(), 1, nb, eigv.as<double>(), 1, s, 0.0, nullptr, 0, 0,
                         d_vectors + x0, 1, ldv, gws.as<double>(), gws_elems, 0.0, nullptr));
    hipLaunchKernelGGL(nys_rowscale_kernel, dim3(ceil_div((long)nb * K, 256)), dim3(256), 0, st, d_vectors + x0, nb, K, ldv,
                       fac.as<double>());
    FLGP_TRY(check_launch("nys_rowscale_kernel"));
  }
  FLGP_HIP(hipStreamSynchronize(st));
  return FLGP_OK;
}
