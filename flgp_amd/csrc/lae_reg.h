// Register-resident Local Anchor Embedding kernels.
//
// lae_kernel (lae.hip) keeps each point's r gathered anchors in LDS; every residual evaluation then
// streams r*d doubles through the LDS pipe, which feeds one wave-wide 8-byte read per 4 clocks per CU
// while the four SIMDs could retire four fp64 FMAs in the same time -- and for r*d >= 160 the 80+ KB per
// wave leave two waves (or fewer lanes) per CU.  Here the anchors live in VGPRs: LP lanes share one
// point, lane `sub` holding coordinates [sub*DPL, (sub+1)*DPL) of all r anchors (R x DPL doubles), so
// the r x d products of a residual evaluation touch no memory.  Only the packed Gram block (LDS, one copy
// per point) and the point itself (LDS) are read from memory inside the iteration.
//
// Arithmetic is the oracle's, operation for operation (local_anchor_embedding_cpp, reference
// src/lae.cpp:83-132): every k-ascending FMA chain that crosses lanes is continued in coordinate order --
// lane 0 runs its DPL terms, hands the partial sum to lane 1 by DPP quad-permute, and so on; all lanes of
// a group execute every leg and keep the leg owner's result, so the r-vectors stay replicated bit for
// bit and the simplex projection / line search need no further exchange.
#pragma once
#include "common.h"
#include "lae_dev.h"

namespace flgp {

// broadcast the value held by lane `j` of each LP-lane group to the whole group (LP in {1, 2, 4})
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int LP>
__device__ __forceinline__ double group_bcast(double v, int j) {
  if constexpr (LP == 1) {
    return v;
  } else if constexpr (LP == 2) {
    return j == 0 ? dpp_quad<0xA0>(v) : dpp_quad<0xF5>(v);          // quad_perm [0,0,2,2] / [1,1,3,3]
  } else {
    static_assert(LP == 4, "LP must be 1, 2 or 4");
    switch (j) {
      case 0: return dpp_quad<0x00>(v);
      case 1: return dpp_quad<0x55>(v);
      case 2: return dpp_quad<0xAA>(v);
      default: return dpp_quad<0xFF>(v);
    }
  }
}

// Iteration-budget passes (round 3).  The iteration count is data dependent -- mean 10.4 per point at BASELINE
// configs[2], mean of the per-wave maximum 20 -- and a wave runs until its slowest point has converged: half of the lane
// cycles are spent on points that are done.  So the work is cut in two: pass 1 gives every point `t_cut` iterations; a
// point that has not met the stop test by then parks its state (z_prev, z_curr, the exponent of beta: 2r + 1 doubles; the
// momentum sequence is the same for every point and indexed by t) in a list, and pass 2 runs over the COMPACTED list
// from iteration t_cut on.  The iteration is deterministic in that state, so every point goes through exactly the
// operations it went through before: bit for bit the same weights (tests/test_gpu_parity.py::test_lae_bit_exact).
struct LaeCont {
  int phase;            // 0: one pass (all 100 iterations); 1: first pass, park the unfinished; 2: continuation over the list
  int t_cut;
  int cap;              // capacity of the list (= n)
  int *count;           // number of parked points
  int *list;            // their indices
  double *state;        // [2R + 1][cap]
};

template <int R, int DPL, int LP>
__global__ __launch_bounds__(64) void lae_reg_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                     const double *__restrict__ Ut, int dpad,
                                                     const int *__restrict__ knn_idx, int ldk,
                                                     int *__restrict__ ell_idx, double *__restrict__ ell_val,
                                                     LaeMomentum mom, LaeCont ct) {
  constexpr int PTS = 64 / LP;
  constexpr int NG = R * (R + 1) / 2;
  __shared__ double Gl[NG * PTS];    // packed upper triangle of U_i U_i^T, [e][point]
  __shared__ double Xl[DPL * 64];    // this lane's slice of the point, [k][lane]
  const int tid = threadIdx.x;
  const int sub = tid & (LP - 1), pl = tid / LP;
  const int kb = sub * DPL;
  long i = (long)blockIdx.x * PTS + pl;
  bool live = i < n;
  long slot = 0;
  if (ct.phase == 2) {
    const int cnt = *ct.count;
    if ((long)blockIdx.x * PTS >= cnt) return;          // (uniform: the whole wave)
    slot = i;
    live = slot < cnt;
    if (!live) slot = cnt - 1;                           // idle lanes shadow the last parked point: no extra iterations
    i = ct.list[slot];
  }
  if (!live && ct.phase != 2) i = n - 1;
  int id[R];
#pragma unroll
  for (int a = 0; a < R; ++a) id[a] = knn_idx[(size_t)a * ldk + i];
  double u[R][DPL];
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int k = 0; k < DPL; ++k) u[a][k] = (kb + k < d) ? Ut[(size_t)id[a] * dpad + kb + k] : 0.0;
#pragma unroll
  for (int k = 0; k < DPL; ++k) Xl[k * 64 + tid] = (kb + k < d) ? X[(size_t)(kb + k) * ldx + i] : 0.0;

  // sum_k p(k) q(k) over all coordinates as ONE k-ascending chain (first term a product, the rest FMAs),
  // continued lane to lane.  Coordinates k >= d are zero in both operands: fma(0, 0, acc) returns acc
  // (up to the sign of a zero, which no comparison or non-zero value downstream can see), so the padded
  // chain ends on the oracle's d-term value.
  auto chain = [&](auto term) -> double {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      double t = acc;
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        double p, q;
        term(k, p, q);
        t = (j == 0 && k == 0) ? p * q : __builtin_fma(p, q, t);
      }
      acc = group_bcast<LP>(t, j);
    }
    return acc;
  };

  // UUt (src/lae.cpp:90) and x*Ut
  double xUt[R];
  {
    int e = 0;
#pragma unroll
    for (int a = 0; a < R; ++a) {
      xUt[a] = chain([&](int k, double &p, double &q) { p = Xl[k * 64 + tid]; q = u[a][k]; });
#pragma unroll
      for (int b = a; b < R; ++b, ++e) {
        const double g = chain([&](int k, double &p, double &q) { p = u[a][k]; q = u[b][k]; });
        if (sub == 0) Gl[e * PTS + pl] = g;
      }
    }
  }
  // Lane `sub == 0` of a group stored the block, the other lanes of the group read it: the hardware completes one
  // wave's LDS operations in order, but without a fence the COMPILER may move those loads above the (to it,
  // unrelated and conditional) stores -- it did for r <= 5 with two or four lanes per point, where everything is
  // unrolled into registers: the lanes that did not store read stale LDS (scripts/sweep_lae.py).  The workgroup is
  // this single wave, so the barrier costs nothing at run time.
  __syncthreads();
  auto Gab = [&](int a, int b) -> double {
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return Gl[(lo * R - lo * (lo - 1) / 2 + (hi - lo)) * PTS + pl];
  };
  auto half_sq_resid = [&](const double *zz) -> double {
    if constexpr (LP == 1) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        double zu = zz[0] * u[0][k];
#pragma unroll
        for (int a = 1; a < R; ++a) zu = __builtin_fma(zz[a], u[a][k], zu);
        const double df = Xl[k * 64 + tid] - zu;
        acc = (k == 0) ? df * df : __builtin_fma(df, df, acc);
      }
      return acc / 2.0;
    } else {
      double df[DPL];
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        double zu = zz[0] * u[0][k];
#pragma unroll
        for (int a = 1; a < R; ++a) zu = __builtin_fma(zz[a], u[a][k], zu);
        df[k] = Xl[k * 64 + tid] - zu;
      }
      return chain([&](int k, double &p, double &q) { p = df[k]; q = df[k]; }) / 2.0;
    }
  };

  double zp[R], zc[R], v[R], grad[R], z[R];
  const double z0 = 1.0 / (double)R;
#pragma unroll
  for (int a = 0; a < R; ++a) { zp[a] = z0; zc[a] = z0; }
  int be = 0;  // beta_curr = 2^be
  if (ct.phase == 2) {
#pragma unroll
    for (int a = 0; a < R; ++a) { zp[a] = ct.state[(size_t)a * ct.cap + slot]; zc[a] = ct.state[(size_t)(R + a) * ct.cap + slot]; }
    be = (int)ct.state[(size_t)(2 * R) * ct.cap + slot];
  }
  const int t_begin = ct.phase == 2 ? ct.t_cut : 0, t_end = ct.phase == 1 ? ct.t_cut : 100;
  bool done = false;
  for (int t = t_begin; t < t_end; ++t) {
    const double alpha = mom.alpha[t];
#pragma unroll
    for (int a = 0; a < R; ++a) v[a] = zc[a] + alpha * (zc[a] - zp[a]);
    const double g_v = half_sq_resid(v);
    // grad = G v - x U^T with every entry an ascending chain over b (src/lae.cpp:104), from ONE pass over the upper triangle of
    // the symmetric block: entry (a, b) is read once and feeds grad[b] (source a) and, for b > a, grad[a] (source b).  Row by
    // row this hands every target its sources in ascending order -- a < t at rows a, then t itself, then b > t along row t --
    // so the chains are the oracle's, with 55 LDS reads instead of 100 at r = 10 (the LDS pipe is what four such waves per
    // CU share).
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
      for (int b = a; b < R; ++b) {
        const double g = Gab(a, b);
        grad[b] = (a == 0) ? v[0] * g : __builtin_fma(v[a], g, grad[b]);          // source a into target b (a <= b)
        if (b > a) grad[a] = __builtin_fma(v[b], g, grad[a]);                     // source b into target a (after its own row's diagonal)
      }
    }
#pragma unroll
    for (int a = 0; a < R; ++a) grad[a] = grad[a] - xUt[a];
    for (int j = 0;; ++j) {
      const double beta = pow2(be + j);
      const double ib = inv_pow2(be + j);
      double vt[R];
#pragma unroll
      for (int a = 0; a < R; ++a) vt[a] = v[a] - ib * grad[a];
      v_to_z_dev<R>(vt, z, R);
      const double g_z = half_sq_resid(z);
      double gd = 0.0, sq = 0.0;
#pragma unroll
      for (int a = 0; a < R; ++a) {
        const double dz = z[a] - v[a];
        gd = (a == 0) ? grad[0] * dz : __builtin_fma(grad[a], dz, gd);
        sq = (a == 0) ? dz * dz : __builtin_fma(dz, dz, sq);
      }
      const double g_t = (g_v + gd) + (beta * sq) / 2.0;
      if (g_z <= g_t || j >= 64) {
        be += j;
#pragma unroll
        for (int a = 0; a < R; ++a) { zp[a] = zc[a]; zc[a] = z[a]; }
        break;
      }
    }
    double sq = 0.0;
#pragma unroll
    for (int a = 0; a < R; ++a) {
      const double df = zc[a] - zp[a];
      sq = (a == 0) ? df * df : __builtin_fma(df, df, sq);
    }
    if (sq < 1e-5) { done = true; break; }
  }
  if (ct.phase == 1) {
    // park the points that are not done: one counter increment per wave, slots in lane order
    const bool park = live && !done && sub == 0;
    const unsigned long long mask = __ballot(park);
    int base = 0;
    if (mask) {
      if (tid == (int)__builtin_ctzll(mask)) base = atomicAdd(ct.count, (int)__builtin_popcountll(mask));
      base = __shfl(base, (int)__builtin_ctzll(mask), 64);
    }
    if (park) {
      const long sl = base + (long)__builtin_popcountll(mask & ((1ull << tid) - 1ull));
      ct.list[sl] = (int)i;
#pragma unroll
      for (int a = 0; a < R; ++a) { ct.state[(size_t)a * ct.cap + sl] = zp[a]; ct.state[(size_t)(R + a) * ct.cap + sl] = zc[a]; }
      ct.state[(size_t)(2 * R) * ct.cap + sl] = (double)be;
    }
    if (!done) live = false;       // its row is written by the second pass
  }
  // ELL row sorted by anchor index (what the CSR conversion of the reference produces)
  int key[R];
#pragma unroll
  for (int a = 0; a < R; ++a) key[a] = id[a];
#pragma unroll
  for (int pass = 0; pass < R; ++pass) {
#pragma unroll
    for (int a = pass & 1; a + 1 < R; a += 2) {
      const bool sw = key[a + 1] < key[a];
      const int k0 = sw ? key[a + 1] : key[a], k1 = sw ? key[a] : key[a + 1];
      const double w0 = sw ? zc[a + 1] : zc[a], w1 = sw ? zc[a] : zc[a + 1];
      key[a] = k0; key[a + 1] = k1; zc[a] = w0; zc[a + 1] = w1;
    }
  }
  if (live && sub == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) {
      ell_idx[(size_t)i * R + a] = key[a];
      ell_val[(size_t)i * R + a] = zc[a];
    }
  }
}

#define FLGP_LAE_REG_ARGS                                                                                      \
  hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad, const int *d_knn, int ldk, \
      int *d_ei, double *d_ev

template <int R, int DPL, int LP>
int launch_lae_reg_t(FLGP_LAE_REG_ARGS) {
  ProfScope ps("lae_kernel", st, 8.0 * (double)n * R);
  const int t_cut = tuning("lae_cut", 13);   // (C3: one pass 2.62 ms; cut 8 / 10 / 12 / 14 / 16 / 18: 2.82 / 2.58 / 2.44 / 2.43 / 2.54 / 2.60)
  if (t_cut <= 0 || t_cut >= 100 || n < tuning("lae_cut_min_n", 32768)) {
    hipLaunchKernelGGL((lae_reg_kernel<R, DPL, LP>), dim3(ceil_div(n, 64 / LP)), dim3(64), 0, st, dX, n, ldx, d, dUt,
                       dpad, d_knn, ldk, d_ei, d_ev, lae_momentum(), LaeCont{0, 0, 0, nullptr, nullptr, nullptr});
    return check_launch("lae_reg_kernel");
  }
  // two passes with the unfinished points compacted in between (see LaeCont)
  DevBuf list, state;
  FLGP_TRY(list.alloc(sizeof(int) * ((size_t)n + 64)));
  FLGP_TRY(state.alloc(sizeof(double) * (size_t)n * (2 * R + 1)));
  int *count = list.as<int>() + n;          // (the counter lives behind the list)
  FLGP_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
  LaeCont ct{1, t_cut, n, count, list.as<int>(), state.as<double>()};
  hipLaunchKernelGGL((lae_reg_kernel<R, DPL, LP>), dim3(ceil_div(n, 64 / LP)), dim3(64), 0, st, dX, n, ldx, d, dUt,
                     dpad, d_knn, ldk, d_ei, d_ev, lae_momentum(), ct);
  ct.phase = 2;
  hipLaunchKernelGGL((lae_reg_kernel<R, DPL, LP>), dim3(ceil_div(n, 64 / LP)), dim3(64), 0, st, dX, n, ldx, d, dUt,
                     dpad, d_knn, ldk, d_ei, d_ev, lae_momentum(), ct);
  return check_launch("lae_reg_kernel");     // (list / state go back to the cache: that synchronises the device)
}

#define FLGP_LAE_REG_PASS st, dX, n, ldx, d, dUt, dpad, d_knn, ldk, d_ei, d_ev

// Kernels built for a compile-time r.  One lane per point on the narrowest slice that holds d; once
// R x 16 doubles no longer fit the 256 architectural VGPRs (R >= 9: the rest would sit in AGPRs behind
// two v_accvgpr_read per operand) two lanes share a point.  Measured on 1e6 points, r = 10, d = 16:
// 16x1 3.0 ms, 8x2 2.6 ms, 4x4 4.0 ms; LDS-resident kernel 8.5 ms (scripts/exp_lae.py).
template <int R>
int launch_lae_reg_r(FLGP_LAE_REG_ARGS, int force_dpl, int force_lp) {
  if (force_dpl == 4 && force_lp == 4 && d <= 16) return launch_lae_reg_t<R, 4, 4>(FLGP_LAE_REG_PASS);
  if (force_dpl == 8 && force_lp == 4 && d <= 32) return launch_lae_reg_t<R, 8, 4>(FLGP_LAE_REG_PASS);
  if (force_dpl == 8 && force_lp == 2 && d <= 16) return launch_lae_reg_t<R, 8, 2>(FLGP_LAE_REG_PASS);
  if (d <= 4) return launch_lae_reg_t<R, 4, 1>(FLGP_LAE_REG_PASS);
  if (d <= 8) return launch_lae_reg_t<R, 8, 1>(FLGP_LAE_REG_PASS);
  if constexpr (R <= 10) {
    if (d <= 16 && (R <= 8 || force_dpl == 16)) return launch_lae_reg_t<R, 16, 1>(FLGP_LAE_REG_PASS);
    if (d <= 16) return launch_lae_reg_t<R, 8, 2>(FLGP_LAE_REG_PASS);
    if (d <= 32) return launch_lae_reg_t<R, 16, 2>(FLGP_LAE_REG_PASS);
    if (d <= 64) return launch_lae_reg_t<R, 16, 4>(FLGP_LAE_REG_PASS);
  } else {
    if (d <= 16) return launch_lae_reg_t<R, 8, 2>(FLGP_LAE_REG_PASS);
    if (d <= 32) return launch_lae_reg_t<R, 8, 4>(FLGP_LAE_REG_PASS);
  }
  return FLGP_LAE_REG_NONE;
}

}  // namespace flgp
