// The K x K (or m x m) dense algebra of the regression consumers of an EigenPair, on the device:
// Cholesky factorisation + solves of the Woodbury system in predict_regression_cpp (reference src/Predict.cpp:40-75) and
// posterior_covariance_regression (src/Utils.cpp:214-250), so that only m_new-length results cross PCIe.
// The systems are small (K <= a few hundred: K = 100 at BASELINE configs[1]) and strictly sequential in their outer
// loop, so one workgroup does each: nothing here is near a roofline, it only has to stay off the host.
#include "common.h"

namespace flgp {

// In-place lower Cholesky factor of the SPD N x N matrix A (column-major, lda = N), then B <- A^-1 B for the nrhs
// columns of B (column-major, ldb = N).  One workgroup.  flag[0] is set when a pivot is not positive.
__global__ __launch_bounds__(1024) void chol_solve_kernel(double *__restrict__ A, int N, double *__restrict__ B, int nrhs,
                                                          int *__restrict__ flag) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int j = 0; j < N; ++j) {
    if (tid == 0) {
      const double d = A[(size_t)j * N + j];
      if (!(d > 0.0)) bad = 1;
      A[(size_t)j * N + j] = __builtin_sqrt(d > 0.0 ? d : 1.0);
    }
    __syncthreads();
    if (bad) break;
    const double inv = 1.0 / A[(size_t)j * N + j];
    for (int i = j + 1 + tid; i < N; i += 1024) A[(size_t)j * N + i] *= inv;
    __syncthreads();
    // trailing update of the lower triangle: A(i, k) -= L(i, j) L(k, j), i >= k > j; a wave per column
    for (int k = j + 1 + wave; k < N; k += 16) {
      const double lkj = A[(size_t)j * N + k];
      for (int i = k + lane; i < N; i += 64) A[(size_t)k * N + i] -= A[(size_t)j * N + i] * lkj;
    }
    __syncthreads();
  }
  if (bad && tid == 0) flag[0] = 1;
}

// B <- A^-1 B with the factor chol_solve_kernel left in A: L y = b, then L^T x = y, a thread per right-hand side, the
// right-hand sides spread over the whole grid.  (Round 2 solved them inside the factorisation's one workgroup:
// posterior_covariance_regression with m <= K passes m_new right-hand sides, and m_new ~ n = 1e6 of them on a single CU
// took minutes -- ADVICE r02.  The arithmetic per right-hand side is unchanged, so are the bits.)
__global__ __launch_bounds__(128) void chol_apply_kernel(const double *__restrict__ A, int N, double *__restrict__ B, int nrhs,
                                                         const int *__restrict__ flag) {
  if (flag[0]) return;
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrhs) return;
  double *b = B + (size_t)c * N;
  for (int i = 0; i < N; ++i) {
    double acc = b[i];
    for (int k = 0; k < i; ++k) acc -= A[(size_t)k * N + i] * b[k];
    b[i] = acc / A[(size_t)i * N + i];
  }
  for (int i = N - 1; i >= 0; --i) {
    double acc = b[i];
    const double *li = A + (size_t)i * N;       // column i of L = row i of L^T
    for (int k = i + 1; k < N; ++k) acc -= li[k] * b[k];
    b[i] = acc / li[i];
  }
}

int chol_solve(hipStream_t st, double *dA, int N, double *dB, int nrhs, int *d_flag) {
  hipLaunchKernelGGL(chol_solve_kernel, dim3(1), dim3(1024), 0, st, dA, N, dB, nrhs, d_flag);
  FLGP_TRY(check_launch("chol_solve_kernel"));
  if (nrhs > 0) {
    hipLaunchKernelGGL(chol_apply_kernel, dim3(ceil_div(nrhs, 128)), dim3(128), 0, st, dA, N, dB, nrhs, d_flag);
    FLGP_TRY(check_launch("chol_apply_kernel"));
  }
  return FLGP_OK;
}

// lam_k = 1 - values_k ; ls = exp(-t lam / 2), l = exp(-t lam)      (src/Predict.cpp:60,64; src/Utils.cpp:220,224)
__global__ void gpr_weights_kernel(const double *__restrict__ values, int K, double t, double *__restrict__ ls,
                                   double *__restrict__ l) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double lam = 1.0 - values[k];
  ls[k] = exp(-0.5 * t * lam) + 0.0;
  l[k] = exp(-t * lam);
}

// Q = diag(ls) VtV diag(ls) + c I                                    (src/Predict.cpp:65-66)
__global__ void gpr_q_kernel(const double *__restrict__ VtV, const double *__restrict__ ls, int K, double c,
                             double *__restrict__ Q) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = (int)(e % K), j = (int)(e / K);
  Q[e] = ls[i] * VtV[e] * ls[j] + (i == j ? c : 0.0);
}

// out(i, j) = a(i) * M(i, j) * b(j) (either scaling optional), K x q column-major
__global__ void gpr_scale_kernel(const double *__restrict__ M, const double *__restrict__ a, const double *__restrict__ b,
                                 int rows, int cols, double *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  const int i = (int)(e % rows), j = (int)(e / rows);
  double v = M[e];
  if (a) v *= a[i];
  if (b) v *= b[j];
  out[e] = v;
}

// out = alpha * (X - Y)   elementwise
__global__ void gpr_diff_kernel(const double *__restrict__ X, const double *__restrict__ Y, double alpha, long count,
                                double *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) out[e] = alpha * (X[e] - Y[e]);
}

__global__ void gpr_add_diag_kernel(double *__restrict__ A, int N, double c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) A[(size_t)i * N + i] += c;
}
// A(i, i) += v[i]
__global__ void gpr_add_diag_vec_kernel(double *__restrict__ A, int N, const double *__restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) A[(size_t)i * N + i] += v[i];
}
// out(i, j) = a(i) * M(i, j): M rows x cols with leading dimension ldm, out contiguous (rows x cols)
__global__ void gpr_rowscale_ld_kernel(const double *__restrict__ M, long ldm, const double *__restrict__ a, int rows, int cols,
                                       double *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  const int i = (int)(e % rows), j = (int)(e / rows);
  out[e] = a[i] * M[(size_t)j * ldm + i];
}
// zinv[i] = 1 / (noise[i] + sigma)
__global__ void gpr_zinv_kernel(const double *__restrict__ noise, double sigma, int m, double *__restrict__ zinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) zinv[i] = 1.0 / (noise[i] + sigma);
}

// cov_i = sum_k V2(i,k)^2 l_k + c - sum_k V2(i,k) W(i,k)            (src/Utils.cpp:249: rowwise sums, k ascending)
__global__ void gpr_rowquad_kernel(const double *__restrict__ V2, long ld2, const double *__restrict__ W, int mnew, int K,
                                   const double *__restrict__ l, double c, double *__restrict__ cov) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mnew) return;
  double prior = 0.0, beta = 0.0;
  for (int k = 0; k < K; ++k) {
    const double v = V2[(size_t)k * ld2 + i];
    prior += (v * l[k]) * v;
    beta += v * W[(size_t)k * mnew + i];
  }
  cov[i] = prior + c - beta;
}

// m <= K branch (src/Utils.cpp:228-236,249): C12 = HK(idx0, idx1) and X = K11^-1 C12 are m x mnew column-major;
// beta_i = sum_b C21(i,b) alpha(i,b) = sum_b C12(b,i) X(b,i); cov_i = prior_i + c - beta_i
__global__ void gpr_rowdot_kernel(const double *__restrict__ C12, const double *__restrict__ X, int mnew, int m,
                                  const double *__restrict__ V2, long ld2, int K, const double *__restrict__ l, double c,
                                  double *__restrict__ cov) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mnew) return;
  double beta = 0.0, prior = 0.0;
  for (int b = 0; b < m; ++b) beta += C12[(size_t)i * m + b] * X[(size_t)i * m + b];
  for (int k = 0; k < K; ++k) { const double v = V2[(size_t)k * ld2 + i]; prior += (v * l[k]) * v; }
  cov[i] = prior + c - beta;
}

int gpr_weights(hipStream_t st, const double *d_values, int K, double t, double *d_ls, double *d_l) {
  hipLaunchKernelGGL(gpr_weights_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, st, d_values, K, t, d_ls, d_l);
  return check_launch("gpr_weights_kernel");
}
int gpr_q(hipStream_t st, const double *dVtV, const double *d_ls, int K, double c, double *dQ) {
  hipLaunchKernelGGL(gpr_q_kernel, dim3(ceil_div((long)K * K, 256)), dim3(256), 0, st, dVtV, d_ls, K, c, dQ);
  return check_launch("gpr_q_kernel");
}
int gpr_scale(hipStream_t st, const double *dM, const double *d_a, const double *d_b, int rows, int cols, double *d_out) {
  hipLaunchKernelGGL(gpr_scale_kernel, dim3(ceil_div((long)rows * cols, 256)), dim3(256), 0, st, dM, d_a, d_b, rows, cols, d_out);
  return check_launch("gpr_scale_kernel");
}
int gpr_diff(hipStream_t st, const double *dX, const double *dY, double alpha, long count, double *d_out) {
  hipLaunchKernelGGL(gpr_diff_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, st, dX, dY, alpha, count, d_out);
  return check_launch("gpr_diff_kernel");
}
int gpr_add_diag(hipStream_t st, double *dA, int N, double c) {
  hipLaunchKernelGGL(gpr_add_diag_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, dA, N, c);
  return check_launch("gpr_add_diag_kernel");
}
int gpr_add_diag_vec(hipStream_t st, double *dA, int N, const double *d_v) {
  hipLaunchKernelGGL(gpr_add_diag_vec_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, dA, N, d_v);
  return check_launch("gpr_add_diag_vec_kernel");
}
int gpr_rowscale_ld(hipStream_t st, const double *dM, long ldm, const double *d_a, int rows, int cols, double *d_out) {
  hipLaunchKernelGGL(gpr_rowscale_ld_kernel, dim3(ceil_div((long)rows * cols, 256)), dim3(256), 0, st, dM, ldm, d_a, rows, cols, d_out);
  return check_launch("gpr_rowscale_ld_kernel");
}
int gpr_zinv(hipStream_t st, const double *d_noise, double sigma, int m, double *d_zinv) {
  hipLaunchKernelGGL(gpr_zinv_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, d_noise, sigma, m, d_zinv);
  return check_launch("gpr_zinv_kernel");
}
int gpr_rowquad(hipStream_t st, const double *dV2, long ld2, const double *dW, int mnew, int K, const double *d_l, double c,
                double *d_cov) {
  hipLaunchKernelGGL(gpr_rowquad_kernel, dim3(ceil_div(mnew, 256)), dim3(256), 0, st, dV2, ld2, dW, mnew, K, d_l, c, d_cov);
  return check_launch("gpr_rowquad_kernel");
}
int gpr_rowdot(hipStream_t st, const double *dC21, const double *dAl, int mnew, int m, const double *dV2, long ld2, int K,
               const double *d_l, double c, double *d_cov) {
  hipLaunchKernelGGL(gpr_rowdot_kernel, dim3(ceil_div(mnew, 256)), dim3(256), 0, st, dC21, dAl, mnew, m, dV2, ld2, K, d_l, c, d_cov);
  return check_launch("gpr_rowdot_kernel");
}

}  // namespace flgp
