// k3 + k4: Local Anchor Embedding weights and the ELL form of the n x s similarity matrix Z.
//
// Replaces LAE_cpp / LAE_Parallel / local_anchor_embedding_cpp / v_to_z_cpp,
// reference src/lae.cpp:15-153, and the serial Z_sp.insert loop (:60-67).
//
// One lane owns one point and runs the whole Nesterov projected-gradient iteration for it
// (data-dependent trip counts: lanes of a wave simply retire at different times).  The r
// gathered anchors of the point live in LDS, laid out [a][k][lane] so that a lane's read is
// conflict-free; the r x r Gram block U_i U_i^T lives in registers (r <= 10) or LDS.
//
// The arithmetic follows oracle/flgp_oracle.c operation for operation (k-ascending FMA
// chains for dot products, one rounded IEEE operation per source operator otherwise, correctly
// rounded division and sqrt, exact ldexp for 2^j beta), so the weights agree with the oracle
// bit for bit.  Output: for every point its r (anchor, weight) pairs sorted by anchor index,
// i.e. the inner order of the reference's row-major Eigen::SparseMatrix == dgRMatrix slots.
#include "common.h"
#include "lae_dev.h"

namespace flgp {

__global__ void v_to_z_kernel(const double *__restrict__ v, int r, double *__restrict__ z) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double vv[FLGP_RMAX], zz[FLGP_RMAX];
  for (int a = 0; a < FLGP_RMAX; ++a) { vv[a] = (a < r) ? v[a] : 0.0; zz[a] = 0.0; }
  v_to_z_dev<0>(vv, zz, r);
  for (int a = 0; a < r; ++a) z[a] = zz[a];
}

template <int R, bool GREG, bool UI_LDS>
__global__ __launch_bounds__(64) void lae_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                 const double *__restrict__ Ut, int dpad, int r_rt,
                                                 const int *__restrict__ knn_idx, int ldk,
                                                 int *__restrict__ ell_idx, double *__restrict__ ell_val,
                                                 int *__restrict__ iters_out, int NT, LaeMomentum mom) {
  // NT = lanes of the wave that own a point (64 unless the per-point LDS footprint forces fewer)
  constexpr int RR = LaeDims<R>::RR;
  const int r = R ? R : r_rt;
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  long i = (long)blockIdx.x * NT + tid;
  const bool live = tid < NT && i < n;
  if (i >= n) i = n - 1;
  if (tid >= NT) return;   // no barriers in this kernel: idle lanes may leave

  double *Ul = lds;                                         // [a][k][lane]
  double *Gl = lds + (UI_LDS ? (size_t)r * d * NT : 0);     // [a][b][lane] when !GREG
  int id[RR];
#pragma unroll
  for (int a = 0; a < RR; ++a) id[a] = (a < r) ? knn_idx[(size_t)a * ldk + i] : 0;

  auto Uak = [&](int a, int k) -> double {
    if (UI_LDS) return Ul[((size_t)a * d + k) * NT + tid];
    return Ut[(size_t)id[a] * dpad + k];
  };
  auto xk = [&](int k) -> double { return X[(size_t)k * ldx + i]; };

  if (UI_LDS) {
#pragma unroll
    for (int a = 0; a < RR; ++a)
      if (a < r)
        for (int k = 0; k < d; ++k) Ul[((size_t)a * d + k) * NT + tid] = Ut[(size_t)id[a] * dpad + k];
  }

  // UUt (src/lae.cpp:90) and x*Ut: k-ascending FMA chains
  constexpr int NG = GREG ? RR * (RR + 1) / 2 : 1;
  double g[NG];
  double xUt[RR];
  {
    if (GREG) {
#pragma unroll
      for (int e = 0; e < NG; ++e) g[e] = 0.0;
    }
#pragma unroll
    for (int a = 0; a < RR; ++a) xUt[a] = 0.0;
    for (int k = 0; k < d; ++k) {
      double u[RR];
#pragma unroll
      for (int a = 0; a < RR; ++a) u[a] = (a < r) ? Uak(a, k) : 0.0;
      const double xv = xk(k);
      int e = 0;
#pragma unroll
      for (int a = 0; a < RR; ++a) {
        if (a < r) xUt[a] = (k == 0) ? xv * u[a] : __builtin_fma(xv, u[a], xUt[a]);
#pragma unroll
        for (int b = a; b < RR; ++b, ++e) {
          if (GREG) {
            g[e] = (k == 0) ? u[a] * u[b] : __builtin_fma(u[a], u[b], g[e]);
          } else if (b < r) {
            const double old = (k == 0) ? 0.0 : Gl[((size_t)a * r + b) * NT + tid];
            const double nw = (k == 0) ? u[a] * u[b] : __builtin_fma(u[a], u[b], old);
            Gl[((size_t)a * r + b) * NT + tid] = nw;
            Gl[((size_t)b * r + a) * NT + tid] = nw;
          }
        }
      }
    }
  }
  auto Gab = [&](int a, int b) -> double {  // symmetric: the FMA chain commutes operand-wise
    if (GREG) {
      const int lo = a < b ? a : b, hi = a < b ? b : a;
      return g[lo * RR - lo * (lo - 1) / 2 + (hi - lo)];
    }
    return Gl[((size_t)a * r + b) * NT + tid];
  };

  // g(z) = |x - zU|^2 / 2  (src/lae.cpp:104,118)
  auto half_sq_resid = [&](const double *zz) -> double {
    double acc = 0.0;
#pragma unroll 8     // (the anchors may come from memory: eight coordinates' gathers in flight, the chain stays in k order;
                     //  d = 784, r = 3: 13.3 -> 5.9 ms per 70 000 points; 16 is slower again, 8.9 ms)
    for (int k = 0; k < d; ++k) {
      double zu = zz[0] * Uak(0, k);
#pragma unroll
      for (int a = 1; a < RR; ++a)
        if (a < r) zu = __builtin_fma(zz[a], Uak(a, k), zu);
      const double df = xk(k) - zu;
      acc = (k == 0) ? df * df : __builtin_fma(df, df, acc);
    }
    return acc / 2.0;
  };

  auto v_to_z = [&](const double *vv, double *zz) { v_to_z_dev<R>(vv, zz, r); };

  double zp[RR], zc[RR], v[RR], grad[RR], vt[RR], z[RR], dz[RR];
  const double z0 = 1.0 / (double)r;  // (src/lae.cpp:82)
#pragma unroll
  for (int a = 0; a < RR; ++a) { zp[a] = z0; zc[a] = z0; v[a] = 0; grad[a] = 0; vt[a] = 0; z[a] = 0; dz[a] = 0; }
  int be = 0;  // beta_curr = 2^be (:83-84)
  int t = 0;
  for (; t < 100; ++t) {                // T = 100 (:86)
    const double alpha = mom.alpha[t];  // (:99)
#pragma unroll
    for (int a = 0; a < RR; ++a) v[a] = zc[a] + alpha * (zc[a] - zp[a]);  // (:101)
    const double g_v = half_sq_resid(v);                                  // (:103)
#pragma unroll
    for (int a = 0; a < RR; ++a) {  // grad = v*UUt - x*Ut (:105)
      if (a < r) {
        double acc = v[0] * Gab(0, a);
#pragma unroll
        for (int b = 1; b < RR; ++b)
          if (b < r) acc = __builtin_fma(v[b], Gab(b, a), acc);
        grad[a] = acc - xUt[a];
      }
    }
    for (int j = 0;; ++j) {  // backtracking (:107-129); capped at 64 doublings as the oracle
      const double beta = pow2(be + j);  // std::pow(2,j)*beta_curr (:110), exact
      const double ib = inv_pow2(be + j);
#pragma unroll
      for (int a = 0; a < RR; ++a) vt[a] = v[a] - ib * grad[a];  // (:112)
      v_to_z(vt, z);                                             // (:114)
      const double g_z = half_sq_resid(z);                       // (:116)
      double gd = 0.0, sq = 0.0;
#pragma unroll
      for (int a = 0; a < RR; ++a) {
        dz[a] = z[a] - v[a];
        if (a < r) {
          gd = (a == 0) ? grad[0] * dz[0] : __builtin_fma(grad[a], dz[a], gd);
          sq = (a == 0) ? dz[0] * dz[0] : __builtin_fma(dz[a], dz[a], sq);
        }
      }
      const double g_t = (g_v + gd) + (beta * sq) / 2.0;  // (:117)
      if (g_z <= g_t || j >= 64) {
        be += j;
#pragma unroll
        for (int a = 0; a < RR; ++a) { zp[a] = zc[a]; zc[a] = z[a]; }
        break;
      }
    }
    double sq = 0.0;
#pragma unroll
    for (int a = 0; a < RR; ++a) {
      if (a < r) {
        const double df = zc[a] - zp[a];
        sq = (a == 0) ? df * df : __builtin_fma(df, df, sq);
      }
    }
    if (sq < 1e-5) { ++t; break; }  // tol on the SQUARED norm (:130)
  }

  // sort (anchor, weight) pairs by anchor index: CSR inner order (k4)
  int key[RR];
#pragma unroll
  for (int a = 0; a < RR; ++a) key[a] = (a < r) ? id[a] : 0x7fffffff;
#pragma unroll
  for (int pass = 0; pass < RR; ++pass) {
#pragma unroll
    for (int a = pass & 1; a + 1 < RR; a += 2) {
      const bool sw = key[a + 1] < key[a];
      const int k0 = sw ? key[a + 1] : key[a], k1 = sw ? key[a] : key[a + 1];
      const double w0 = sw ? zc[a + 1] : zc[a], w1 = sw ? zc[a] : zc[a + 1];
      key[a] = k0; key[a + 1] = k1; zc[a] = w0; zc[a + 1] = w1;
    }
  }
  if (live) {
#pragma unroll
    for (int a = 0; a < RR; ++a) {
      if (a < r) {
        ell_idx[(size_t)i * r + a] = key[a];
        ell_val[(size_t)i * r + a] = zc[a];
      }
    }
    if (iters_out) iters_out[i] = t;
  }
}

// SE similarity: Z = exp(-dist/(4 eps^2)) on the stored k-NN entries
// (cross_similarity_se_cpp, reference src/Spectrum.cpp:126-132) -> ELL sorted by column.
__global__ void se_weights_kernel(const int *__restrict__ knn_idx, const double *__restrict__ knn_dist, int n,
                                  int ldk, int r, double den, int *__restrict__ ell_idx,
                                  double *__restrict__ ell_val) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // insertion by ascending anchor index straight into the output row (r <= FLGP_RMAX, tiny)
  int *oi = ell_idx + (size_t)i * r;
  double *ov = ell_val + (size_t)i * r;
  for (int a = 0; a < r; ++a) {
    const int j = knn_idx[(size_t)a * ldk + i];
    const double w = exp(-knn_dist[(size_t)a * ldk + i] / den);
    int p = a;
    while (p > 0 && oi[p - 1] > j) { oi[p] = oi[p - 1]; ov[p] = ov[p - 1]; --p; }
    oi[p] = j; ov[p] = w;
  }
}

template <int R, bool GREG>
static int launch_lae(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad,
                      int r, const int *d_knn, int ldk, int *d_ei, double *d_ev, int *d_iters) {
  const size_t g_bytes = GREG ? 0 : sizeof(double) * (size_t)r * r;   // per point
  const size_t u_bytes = sizeof(double) * (size_t)r * d;
  const size_t budget = 150 * 1024;
  ProfScope ps("lae_kernel", st, 8.0 * (double)n * r);
  // prefer the anchors in LDS with all 64 lanes busy; give up lanes before giving up LDS residency
  // only for the Gram block (which has nowhere else to live when r > 10)
  bool ui_lds = (u_bytes + g_bytes) * 64 <= budget;
  int nt = 64;
  if (!ui_lds)
    while (nt > 8 && g_bytes * nt > budget) nt >>= 1;
  FLGP_REQUIRE(ui_lds || g_bytes * nt <= budget, "LAE: r=%d does not fit the LDS budget", r);
  const size_t lds = (ui_lds ? u_bytes + g_bytes : g_bytes) * nt;
  const int grid = ceil_div(n, nt);
  auto kern = ui_lds ? lae_kernel<R, GREG, true> : lae_kernel<R, GREG, false>;
  if (lds > 48 * 1024)
    FLGP_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, dX, n, ldx, d, dUt, dpad, r, d_knn, ldk, d_ei, d_ev,
                     d_iters, nt, lae_momentum());
  return check_launch("lae_kernel");
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_lae(void *stream, const double *dX, int n, int ldx, int d, const double *dUt, int s,
                            int r, const int *d_knn_idx, int ldk, int *d_ell_idx, double *d_ell_val) {
  hipStream_t st = (hipStream_t)stream;
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "LAE: kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(r >= 1 && r <= s && r <= FLGP_RMAX, "LAE: need 1 <= r <= min(s, %d) (r=%d, s=%d)", FLGP_RMAX, r, s);
  FLGP_REQUIRE(ldx >= n && ldk >= n, "LAE: leading dimensions must be >= n");
  if (n == 0) return FLGP_OK;
  int *it = nullptr;
#define LAE_ARGS st, dX, n, ldx, d, dUt, dpad, r, d_knn_idx, ldk, d_ell_idx, d_ell_val, it
  if (tuning("lae_variant", 1) == 1) {
    // register-resident kernels (lae_reg.h) where r x d/lanes fits the VGPR file; LDS kernels otherwise
    const int rc = launch_lae_reg(st, dX, n, ldx, d, dUt, dpad, r, d_knn_idx, ldk, d_ell_idx, d_ell_val);
    if (rc != FLGP_LAE_REG_NONE) return rc;
  }
  switch (r) {
    case 1: return launch_lae<1, true>(LAE_ARGS);
    case 2: return launch_lae<2, true>(LAE_ARGS);
    case 3: return launch_lae<3, true>(LAE_ARGS);
    case 4: return launch_lae<4, true>(LAE_ARGS);
    case 5: return launch_lae<5, true>(LAE_ARGS);
    case 6: return launch_lae<6, true>(LAE_ARGS);
    case 7: return launch_lae<7, true>(LAE_ARGS);
    case 8: return launch_lae<8, true>(LAE_ARGS);
    case 9: return launch_lae<9, true>(LAE_ARGS);
    case 10: return launch_lae<10, true>(LAE_ARGS);
    case 11: return launch_lae<11, false>(LAE_ARGS);
    case 12: return launch_lae<12, false>(LAE_ARGS);
    case 13: return launch_lae<13, false>(LAE_ARGS);
    case 14: return launch_lae<14, false>(LAE_ARGS);
    case 15: return launch_lae<15, false>(LAE_ARGS);
    case 16: return launch_lae<16, false>(LAE_ARGS);
    default: return launch_lae<0, false>(LAE_ARGS);
  }
}

extern "C" int flgp_dev_v_to_z(void *stream, const double *d_v, int r, double *d_z) {
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX, "v_to_z: need 1 <= r <= %d (got %d)", FLGP_RMAX, r);
  hipLaunchKernelGGL(v_to_z_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_v, r, d_z);
  return check_launch("v_to_z_kernel");
}

// fixed-shape two-level sum (deterministic): partial[b] = sum of a 4096-element slab in a fixed tree
__global__ __launch_bounds__(256) void slab_sum_kernel(const double *__restrict__ x, long count, double *__restrict__ partial) {
  __shared__ double red[256];
  const long base = (long)blockIdx.x * 4096;
  double acc = 0.0;
  for (int k = 0; k < 16; ++k) {
    const long e = base + k * 256 + threadIdx.x;
    if (e < count) acc += x[e];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void final_mean_kernel(const double *__restrict__ partial, int nparts, double denom, double *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double acc = 0.0;
    for (int b = 0; b < nparts; ++b) acc += partial[b];
    out[0] = acc / denom;
  }
}

// mean of the n*r stored k-NN distances (distances_sp.coeffs().sum()/(n*r), reference src/Fit.cpp:131):
// d_out[0]; d_work holds ceil(n*r/4096) doubles
extern "C" int flgp_dev_mean(void *stream, const double *d_x, long count, double *d_out, double *d_work) {
  FLGP_REQUIRE(count >= 1, "mean: empty input");
  const int nparts = ceil_div(count, 4096);
  hipLaunchKernelGGL(slab_sum_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, d_x, count, d_work);
  hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_work, nparts, (double)count, d_out);
  return check_launch("mean kernels");
}

// Z = exp(-dist / den) on the stored entries, den given directly (den = a2 * mean(dist) in the fit_se_*
// bandwidth grid, reference src/Fit.cpp:150)
extern "C" int flgp_dev_se_weights_den(void *stream, const int *d_knn_idx, const double *d_knn_dist, int n, int ldk,
                                       int r, double den, int *d_ell_idx, double *d_ell_val) {
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX && ldk >= n, "SE weights: bad r / ldk");
  if (n == 0) return FLGP_OK;
  hipLaunchKernelGGL(se_weights_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, d_knn_idx,
                     d_knn_dist, n, ldk, r, den, d_ell_idx, d_ell_val);
  return check_launch("se_weights_kernel");
}

extern "C" int flgp_dev_se_weights(void *stream, const int *d_knn_idx, const double *d_knn_dist, int n, int ldk,
                                   int r, double epsilon, int *d_ell_idx, double *d_ell_val) {
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX && ldk >= n, "SE weights: bad r / ldk");
  if (n == 0) return FLGP_OK;
  const double den = (4.0 * epsilon) * epsilon;
  hipLaunchKernelGGL(se_weights_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, d_knn_idx,
                     d_knn_dist, n, ldk, r, den, d_ell_idx, d_ell_val);
  return check_launch("se_weights_kernel");
}
