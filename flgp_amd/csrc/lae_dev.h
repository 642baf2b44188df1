// Device helpers shared by the LAE kernels (lae.hip: anchors in LDS; lae_reg*.hip: anchors in VGPRs).
#pragma once
#include "common.h"

namespace flgp {

// R > 0: compile-time r (everything unrolled, vectors in registers); R == 0: run-time r <= FLGP_RMAX.
template <int R>
struct LaeDims {
  static constexpr int RR = R ? R : FLGP_RMAX;
};

// Batcher's merge-exchange sorting network (Knuth 5.2.2, Algorithm M) for N keys, built at compile time:
// 31 comparators for N = 10 where odd-even transposition needs 45.
template <int N>
struct SortNet {
  static constexpr int CAP = N < 2 ? 1 : N * 6;   // >= N ceil(log2 N)^2 / 4 + ... for N <= 32
  int a[CAP], b[CAP];
  int count;
  constexpr SortNet() : a{}, b{}, count(0) {
    if (N >= 2) {
      int t = 0;
      while ((1 << t) < N) ++t;
      for (int p = 1 << (t - 1); p > 0; p >>= 1) {
        int q = 1 << (t - 1), r = 0, d = p;
        while (true) {
          for (int i = 0; i + d < N; ++i)
            if ((i & p) == r) { a[count] = i; b[count] = i + d; ++count; }
          if (q == p) break;
          d = q - p; q >>= 1; r = p;
        }
      }
    }
  }
};
template <int N>
struct SortNetHolder {
  static constexpr SortNet<N> net = SortNet<N>();
};

// The comparators of the network with their indices as CONSTANT EXPRESSIONS (template recursion).  Read in an unrolled loop
// from the constexpr table they were not folded: the compiler kept the table in memory and indexed the register array
// dynamically -- s_set_gpr_idx_on / v_mov / v_mov / s_set_gpr_idx_off around every operand, 124 times per projection, half
// of the instructions of an LAE iteration (round 3, found in the ISA).
template <int N, int C>
__device__ __forceinline__ void sort_net_apply(double (&vd)[N]) {
  constexpr SortNet<N> net = SortNet<N>();
  if constexpr (C < net.count) {
    constexpr int ia = net.a[C], ib = net.b[C];
    const double hi = __builtin_fmax(vd[ia], vd[ib]);
    const double lo = __builtin_fmin(vd[ia], vd[ib]);
    vd[ia] = hi;
    vd[ib] = lo;
    sort_net_apply<N, C + 1>(vd);
  }
}

// x / J for a small integer constant J, correctly rounded: q0 = RN(x * RN(1/J)) is within one ulp,
// r = x - J q0 is exact in an FMA, and RN(q0 + r * RN(1/J)) is the correctly rounded quotient
// (Markstein's theorem) as long as nothing overflows and the quotient is normal -- the caller checks the
// range.  Three multiply-class instructions instead of the ~25 of an IEEE fp64 division.
template <int J>
__device__ __forceinline__ double div_const(double x) {
  if constexpr ((J & (J - 1)) == 0) {
    return x * (1.0 / (double)J);   // exact
  } else {
    constexpr double rj = 1.0 / (double)J;
    const double q0 = x * rj;
    const double rem = __builtin_fma(-(double)J, q0, x);
    return __builtin_fma(rem, rj, q0);
  }
}

// rho = max{ j : v_(j) - (cumsum_j - 1)/j > 0 }; theta = (cumsum_rho - 1)/rho   (src/lae.cpp:143-150)
template <int RR, int A, bool FAST>
__device__ __forceinline__ void simplex_theta(const double *vd, double c, int r, double &theta) {
  if constexpr (A < RR) {
    if (A < r) {
      c = (A == 0) ? vd[0] : c + vd[A];
      const double num = c - 1.0;
      double q;
      if constexpr (FAST) q = div_const<A + 1>(num);
      else q = num / (double)(A + 1);
      if (A == 0 || vd[A] - q > 0) theta = q;
      simplex_theta<RR, A + 1, FAST>(vd, c, r, theta);
    }
  }
}

// Euclidean projection onto the simplex (v_to_z_cpp, reference src/lae.cpp:137-153)
template <int R>
__device__ __forceinline__ void v_to_z_dev(const double *vv, double *zz, int r) {
  constexpr int RR = LaeDims<R>::RR;
  double vd[RR];
#pragma unroll
  for (int a = 0; a < RR; ++a) vd[a] = (a < r) ? vv[a] : -__builtin_inf();
  // descending sort (any correct sort gives the same array of values)
  sort_net_apply<RR, 0>(vd);
  double theta = 0.0;
  if constexpr (R > 0) {
    // |cumsum_j - 1| is 0 or >= 2^-53 (doubles next to 1), so the quotients are never subnormal; the
    // products stay finite while the extreme entries are below 2^900
    const bool safe = __builtin_fmax(__builtin_fabs(vd[0]), __builtin_fabs(vd[RR - 1])) < 0x1p900;
    if (__builtin_expect(safe, 1)) simplex_theta<RR, 0, true>(vd, 0.0, r, theta);
    else simplex_theta<RR, 0, false>(vd, 0.0, r, theta);
  } else {
    simplex_theta<RR, 0, false>(vd, 0.0, r, theta);
  }
#pragma unroll
  for (int a = 0; a < RR; ++a) {
    if (a < r) {
      const double t = vv[a] - theta;
      zz[a] = t > 0.0 ? t : 0.0;
    }
  }
}

// Nesterov momentum coefficients alpha_t = (delta_{t-1} - 1) / delta_t with delta_0 = 0, delta_1 = 1,
// delta_{t+1} = (1 + sqrt(1 + 4 delta_t^2)) / 2  (src/lae.cpp:84,99,127-128): the same for every point,
// so they are computed once on the host (IEEE sqrt and division, as on the device) and handed to the
// kernels by value -- the kernel argument segment is read through the scalar cache.
struct LaeMomentum {
  double alpha[100];
};
inline LaeMomentum lae_momentum() {
  LaeMomentum m;
  double dp = 0.0, dc = 1.0;
  for (int t = 0; t < 100; ++t) {
    m.alpha[t] = (dp - 1.0) / dc;
    dp = dc;
    dc = (1.0 + __builtin_sqrt(1.0 + (4.0 * dc) * dc)) / 2.0;
  }
  return m;
}

// The Lipschitz estimate beta only ever doubles from 1 (src/lae.cpp:83,110,124): beta = 2^e exactly, and
// 1/beta is another exponent shift rather than a division.
__device__ __forceinline__ double pow2(int e) { return __builtin_ldexp(1.0, e); }
__device__ __forceinline__ double inv_pow2(int e) { return e >= 1024 ? 0.0 : __builtin_ldexp(1.0, -e); }

}  // namespace flgp
