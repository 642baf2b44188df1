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
#include "knn_top.h"

namespace flgp {

// rows of the fp64 panel: s rounded up to the largest anchor tile of the kernels (the screen's 256; the others step by 128)
__host__ __device__ inline int anchor_pad_rows(int s) { return (s + 255) / 256 * 256; }

// ---- the screening copy of the anchors (knn_screen_kernel) ----
// Behind the fp64 panel (rows_pad x dpad doubles at Ut, |u|^2 at uu) sits a copy for the matrix-core screen: every
// coordinate split into two bf16 pieces, u = hi + lo + eps with |eps| <= 2^-17 |u|, laid out as v_mfma_f32_32x32x16_bf16
// A operands -- per tile of 32 anchors 2 KB: hi[khalf 2][anchor 32][8], then lo the same -- and two tables of
// accumulator start values, C1_j = -(|u_j|^2 / 2)(1 + mu) and C2_j = -(|u_j|^2 / 2)(1 - mu) (rows_pad floats each, at
// uu + rows_pad).  Anchors the screen has no say about (|u|^2 above 1e30 or not finite) get a zero row, C1 = -3e38 (never
// a bound) and C2 = +3e38 (always a candidate); padding rows C1 = C2 = -3e38.  Written for dpad <= 16 (64 B per
// anchor whatever dpad: two rows of the narrowest fp64 panel); flgp_dev_anchor_rows() hands out room for it.
constexpr float KNN_SCREEN_MU = 1.220703125e-4f;   // 2^-13, see knn_screen_kernel
constexpr float KNN_SCREEN_BIG = 3.0e38f;
__device__ __host__ inline const unsigned short *screen_panel(const double *Ut, int s, int dpad) {
  return (const unsigned short *)(Ut + (size_t)anchor_pad_rows(s) * dpad);
}
__device__ __host__ inline const float *screen_cinit(const double *uu, int s) { return (const float *)(uu + anchor_pad_rows(s)); }

__device__ __forceinline__ unsigned short bf16_rne(float f) {   // finite f well inside the range
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_as_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void bf16_split(double x, unsigned short &hi, unsigned short &lo) {
  hi = bf16_rne((float)x);
  lo = bf16_rne((float)(x - (double)bf16_as_f32(hi)));
}

__global__ void anchor_prep_kernel(const double *__restrict__ U, int s, int s_pad, int ldu, int d, int dpad,
                                   double *__restrict__ Ut, double *__restrict__ uu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s_pad) return;
  const bool screen = dpad <= 16;
  unsigned short *tile = (unsigned short *)(Ut + (size_t)s_pad * dpad) + (size_t)(j >> 5) * 1024 + (j & 31) * 8;
  float *c1 = (float *)(uu + s_pad), *c2 = c1 + s_pad;
  double acc = 0.0;
  if (j >= s) {  // padding rows: never selected (|u|^2 = +inf makes D = +inf, and inf < thr is false)
    for (int k = 0; k < dpad; ++k) Ut[(size_t)j * dpad + k] = 0.0;
    uu[j] = __builtin_inf();
  } else {
    for (int k = 0; k < d; ++k) {
      const double u = U[(size_t)k * ldu + j];
      Ut[(size_t)j * dpad + k] = u;
      acc = (k == 0) ? u * u : __builtin_fma(u, u, acc);
    }
    for (int k = d; k < dpad; ++k) Ut[(size_t)j * dpad + k] = 0.0;
    uu[j] = acc;
  }
  if (!screen) return;
  const bool usable = j < s && acc <= 1e30;       // false for NaN
  for (int k = 0; k < 16; ++k) {
    unsigned short hi = 0, lo = 0;
    if (usable && k < d) bf16_split(U[(size_t)k * ldu + j], hi, lo);
    const int at = (k >> 3) * 256 + (k & 7);
    tile[at] = hi;
    tile[512 + at] = lo;
  }
  c1[j] = usable ? (float)(-0.5 * acc * (1.0 + (double)KNN_SCREEN_MU)) : -KNN_SCREEN_BIG;
  c2[j] = usable ? (float)(-0.5 * acc * (1.0 - (double)KNN_SCREEN_MU)) : (j < s ? KNN_SCREEN_BIG : -KNN_SCREEN_BIG);
}

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
  constexpr bool NN1 = RCAP == 1;     // r = 1 (Lloyd assignment, cluster counts): a running minimum, no queue
  __shared__ double q_d[NN1 ? 1 : QC * P * NT];
  __shared__ int q_j[NN1 ? 1 : QC * P * NT];
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
          if constexpr (NN1) {
            const bool c = D < top[p].bd[0];            // strict: the lower index keeps a tie
            top[p].bd[0] = c ? D : top[p].bd[0];
            top[p].bi[0] = c ? j0 + jj + a : top[p].bi[0];
          } else if (D < top[p].thr()) {
            q_d[(cnt[p] * P + p) * NT + tid] = D;
            q_j[(cnt[p] * P + p) * NT + tid] = j0 + jj + a;
            ++cnt[p];
          }
        }
      }
      if constexpr (!NN1) {
#pragma unroll
        for (int p = 0; p < P; ++p) full |= (cnt[p] > QC - A);
        if (__any(full)) drain();
      }
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
          // (a row of NaN / Inf coordinates never fills its list: the sentinel must not leave as an index -- every later
          //  stage uses it as an address.  Such a row gets the anchors 0..r-1; the host entry points reject the input.)
          const int bj_ = top[p].bi[k];
          idx_out[(size_t)slot * ldo + i] = ((unsigned)bj_ < (unsigned)s) ? bj_ : slot;
          if (dist_out) dist_out[(size_t)slot * ldo + i] = top[p].bd[k];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k-NN with a matrix-core screen (round 2; d <= 16, r <= 16, 512 <= s <= 32768: a queue entry holds anchor / 8 in twelve bits).
//
// The r nearest anchors and their distances stay EXACT -- the oracle's fp64 chain, bit for bit -- but that chain is only
// run for the dozen anchors per point that can matter.  The rest is decided by E^_j, a bf16x3 matrix-core evaluation of
//     E_j = x . u_j - |u_j|^2 / 2          (D_j = |x|^2 - 2 E_j: larger E = nearer),
// x = hi + lo + eps in bf16 pieces and likewise u; one v_mfma_f32_32x32x16_bf16 each for hi.hi, hi.lo and lo.hi on top of
// an accumulator that starts at -|u_j|^2 / 2: 32 anchors x 32 points and all 16 coordinates per instruction.
//   Error: dropping eps and lo.lo costs <= 3 * 2^-17 sum|x_k u_k|; 49 fp32 accumulations of exact bf16 products cost
//   <= 50 * 2^-22 (sum|x_k u_k| + |u|^2 / 2) even if the hardware truncates; sum|x_k u_k| <= (|x|^2 + |u|^2) / 2; the oracle's
//   own fp64 rounding is 2^-29 of that.  Together |E^ - E| < 2^-15 (|x|^2 + |u|^2) / 2.  The screen budgets four times
//   that: eta_j = mu (|x|^2 + |u_j|^2) / 2, mu = 2^-13, its |u_j|^2 half folded into the start values (C1, C2 of the panel),
//   its |x|^2 half into the point's threshold.
//   Pass 1 (start C1): every lane keeps the maximum of each of its 16 accumulator slots over all tiles: 32 groups of
//   anchors per point (two lanes share a point), each maximum a LOWER bound on some E_j once mu |x|^2 / 2 is taken off.
//   The r-th largest of the 32 is a bound r distinct anchors beat: tau.
//   Pass 2 (start C2): anchor j can only be among the r nearest if E^_j + eta_j >= tau; everything else has r anchors
//   STRICTLY nearer and is dropped.  At s = 5000, r = 10 about 13 anchors per point are left.
//   Then one lane per point runs the exact chain on its candidates (anchor row gathered from the fp64 panel) into a
//   list ordered by (distance, index) -- what the oracle's ascending scan with strict '<' ends with.
// Where the screen has no say the exact arithmetic decides alone: points with |x|^2 outside [1e-30, 1e30] or not finite,
// and points whose candidate queue overflows (more than 32 per half: duplicates of one anchor, say), are scanned against
// all anchors by their whole wave, exactly; anchors with |u|^2 above 1e30 are candidates for every point.
// ------------------------------------------------------------------------------------------
typedef __bf16 kbf8 __attribute__((ext_vector_type(8)));
typedef float kf16 __attribute__((ext_vector_type(16)));
typedef float kf4 __attribute__((ext_vector_type(4)));
typedef float kf2 __attribute__((ext_vector_type(2)));
constexpr int KNN_SCREEN_QC = 32;         // candidates per (point, half)

template <int N>
__device__ __forceinline__ void sort_desc(float (&v)[N]) {   // bitonic network, all indices static
#pragma unroll
  for (int k = 2; k <= N; k *= 2) {
#pragma unroll
    for (int j = k / 2; j > 0; j /= 2) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const float a = v[i], b = v[l];
          const bool desc = (i & k) == 0;
          v[i] = desc ? fmaxf(a, b) : fminf(a, b);
          v[l] = desc ? fminf(a, b) : fmaxf(a, b);
        }
      }
    }
  }
}

// One sweep of the screen over all anchors.  PASS 0 keeps the running maximum of every accumulator slot (g), PASS 1
// queues the anchors at or above the point's threshold.  Per tile of 32 anchors and per 32-point half t of the wave:
// three MFMAs, the two halves' chains interleaved, the next tile's operands on their way from LDS meanwhile.
// The VALU side works on packed differences: acc + 0 (PASS 0) makes the value a known-canonical float for one v_max_f32
// per slot; acc - tau (PASS 1) turns "at or above the threshold" into a clear sign bit, so four slots are ruled out by
// one v_max_i32 + v_max3_i32 on the bit patterns.
template <int PASS, int CH, int QC, int PB>
__device__ __forceinline__ void screen_sweep(const uint4 *__restrict__ panel, const float *__restrict__ ct, int nch,
                                             uint4 (&abuf)[2][CH * 4], float (&cbuf)[2][CH], unsigned short *q,
                                             const kbf8 (&bhi)[2], const kbf8 (&blo)[2], float (&g)[2][16],
                                             const float (&tau)[2], int (&cnt)[2], int tid) {
  constexpr int NT = 256, TPC = CH / 32, PER = CH * 4 / NT;
  const int l = tid & 63, kh = l >> 5, w = tid >> 6, col = l & 31;
  // chunk c of the panel and of the start values into buffer b, by LDS-DMA: each wave-instruction lands 64 x 16 B
  // contiguously at its (wave-uniform) destination, which is exactly the panel's order; no registers held meanwhile
  auto fetch = [&](int c, int b) {
#pragma unroll
    for (int e = 0; e < PER; ++e)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(panel + (size_t)c * (CH * 4) + e * NT + tid),
                                       (__attribute__((address_space(3))) void *)&abuf[b][e * NT + w * 64], 16, 0, 0);
    if (w == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ct + (size_t)c * CH + 4 * l),
                                       (__attribute__((address_space(3))) void *)&cbuf[b][0], 16, 0, 0);
  };
  auto operands = [&](int b, int tile, kbf8 &h, kbf8 &lo, kf16 &Cv) {
    h = __builtin_bit_cast(kbf8, abuf[b][tile * 128 + l]);
    lo = __builtin_bit_cast(kbf8, abuf[b][tile * 128 + 64 + l]);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const kf4 c4 = *(const kf4 *)&cbuf[b][tile * 32 + 8 * g4 + 4 * kh];
      Cv[4 * g4 + 0] = c4[0]; Cv[4 * g4 + 1] = c4[1]; Cv[4 * g4 + 2] = c4[2]; Cv[4 * g4 + 3] = c4[3];
    }
  };
  auto products = [&](const kbf8 &h, const kbf8 &lo, const kf16 &Cv, kf16 (&acc)[2]) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, bhi[0], Cv, 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, bhi[1], Cv, 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, blo[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, blo[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, bhi[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, bhi[1], acc[1], 0, 0, 0);
  };
  auto digest = [&](const kf16 (&acc)[2], int j0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (PASS == 0) {
        // (plain fmaxf: the `+ 0.0f` that stood here until round 3, to hand the compiler a canonical float, had become an
        //  extra v_add per product with this compiler -- 64 of the 128 VALU instructions of a pair of tiles; without it the
        //  maxima fuse into v_max3_f32 and nothing is added)
#pragma unroll
        for (int i = 0; i < 16; ++i) g[t][i] = fmaxf(g[t][i], acc[t][i]);
      } else {
        char *qp = (char *)(q + kh * PB + w * 64 + t * 32 + col);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // a >= tau decided on the group's maximum first (two max instructions and a compare for four products; the
          // difference a - tau per product, as it stood until round 3, was four more -- and a - tau >= 0 iff a >= tau:
          // a difference of two floats is zero only if they are equal, and rounding keeps its sign).  No NaNs here: points
          // the screen has no say about carry tau = +inf and zeroed operands, the panel is finite.
          const float m = fmaxf(fmaxf(fmaxf(acc[t][4 * g4], acc[t][4 * g4 + 1]), acc[t][4 * g4 + 2]), acc[t][4 * g4 + 3]);   // v_max3 + v_max
          if (__ballot(m >= tau[t])) {
            // ONE queue entry per group of four anchors: (group << 4) | which of the four passed.  (Until round 4 every
            // candidate was an entry of its own: four compare / mask / store / count sequences on this path, which half of
            // all groups of a wave enter.)  group = anchor / 8; the half kh of the queue says which four of the eight.
            const unsigned mk = (acc[t][4 * g4] >= tau[t] ? 1u : 0u) | (acc[t][4 * g4 + 1] >= tau[t] ? 2u : 0u) |
                                (acc[t][4 * g4 + 2] >= tau[t] ? 4u : 0u) | (acc[t][4 * g4 + 3] >= tau[t] ? 8u : 0u);
            if (mk) {                                     // cnt counts in bytes of queue stride; past QC entries: overwritten, and counted
              *(unsigned short *)(qp + (cnt[t] & ((QC - 1) * 4 * PB))) = (unsigned short)((((unsigned)j0 >> 3) + g4) << 4 | mk);
              cnt[t] += 4 * PB;
            }
          }
        }
      }
    }
  };
  __syncthreads();            // whoever used the shared buffers before is done with them
  fetch(0, 0);
  __syncthreads();            // (waits for the DMA, then the barrier)
  for (int c = 0; c < nch; ++c) {
    const int b = c & 1;
    if (c + 1 < nch) fetch(c + 1, b ^ 1);     // the other buffer: everyone passed the barrier behind its last reader
    // tile k + 1 multiplies while tile k is digested; the operands of tile k + 2 are on their way from LDS
    kbf8 h0, l0, h1, l1;
    kf16 C0, C1;
    kf16 accA[2], accB[2];
    operands(b, 0, h0, l0, C0);
    operands(b, 1, h1, l1, C1);
    products(h0, l0, C0, accA);
#pragma unroll 1
    for (int tile = 0; tile < TPC; tile += 2) {
      const int j0 = c * CH + tile * 32 + 4 * kh;
      operands(b, tile + 2 < TPC ? tile + 2 : tile, h0, l0, C0);
      products(h1, l1, C1, accB);
      digest(accA, j0);
      operands(b, tile + 3 < TPC ? tile + 3 : tile, h1, l1, C1);
      products(h0, l0, C0, accA);          // past the chunk's last tile: a repeat nobody reads (no branch in the pipeline)
      digest(accB, j0 + 32);
    }
    __syncthreads();
  }
}

template <int DP, int RCAP>
__global__ __launch_bounds__(256, 2) void knn_screen_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                            const double *__restrict__ Ut, const double *__restrict__ uu,
                                                            int s, int r, int *__restrict__ idx_out,
                                                            double *__restrict__ dist_out, int ldo, int stop_after) {
  // stop_after (tuning knob knn_screen_stop, 0 in production): 1 / 2 = return after pass 1 / pass 2 without results, so
  // that scripts/knn_time.py can time the phases of this one kernel
  constexpr int NT = 256, PB = 256, CH = 256, QC = KNN_SCREEN_QC;
  static_assert(DP == 4 || DP == 8 || DP == 16, "the screen panel is written for dpad 4, 8 and 16");
  static_assert(CH == 256, "one 1 KB LDS-DMA moves a chunk's start values");
  static_assert((QC & (QC - 1)) == 0, "queue slots wrap by masking");
  __shared__ uint4 abuf[2][CH * 4];                 // CH anchors x 64 B
  __shared__ __attribute__((aligned(16))) float cbuf[2][CH];
  __shared__ __attribute__((aligned(16))) unsigned short q[QC * 2 * PB];   // [entry][half][point]; first the sorted maxima
  __shared__ int qcnt[2 * PB];
  float *sg = (float *)q;                           // [lane of the workgroup][17]
  static_assert(sizeof(q) >= sizeof(float) * NT * 17, "scratch for the sorted maxima");

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, kh = l >> 5, col = l & 31;
  const long base = (long)blockIdx.x * PB;
  const int s_pad = anchor_pad_rows(s);             // a multiple of 128: the last chunk may be half padding rows ...
  const int nch = (s_pad + CH - 1) / CH;            // ... or, past s_pad, whatever the tables hold behind them: see below
  const uint4 *__restrict__ panel = (const uint4 *)screen_panel(Ut, s, DP);
  const float *__restrict__ ctab = screen_cinit(uu, s);

  // ---- the points as B operands: lane = (point col of half t, coordinates 8 kh .. 8 kh + 7)
  kbf8 bhi[2], blo[2];
  float xxf[2];
  bool off[2];                                      // the screen has no say about this point
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long i = base + w * 64 + t * 32 + col;
    if (i >= n) i = n - 1;
    double part = 0.0;
    unsigned short h[8], lo[8];
    double xv[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = 8 * kh + kk;
      xv[kk] = (k < d) ? X[(size_t)k * ldx + i] : 0.0;
      part = __builtin_fma(xv[kk], xv[kk], part);
    }
    const double xx = part + __shfl_xor(part, 32);
    off[t] = !(xx >= 1e-30 && xx <= 1e30);
    xxf[t] = (float)xx;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      h[kk] = 0; lo[kk] = 0;
      if (!off[t]) bf16_split(xv[kk], h[kk], lo[kk]);
    }
    uint4 ph, pl;
    ph.x = h[0] | ((unsigned)h[1] << 16); ph.y = h[2] | ((unsigned)h[3] << 16);
    ph.z = h[4] | ((unsigned)h[5] << 16); ph.w = h[6] | ((unsigned)h[7] << 16);
    pl.x = lo[0] | ((unsigned)lo[1] << 16); pl.y = lo[2] | ((unsigned)lo[3] << 16);
    pl.z = lo[4] | ((unsigned)lo[5] << 16); pl.w = lo[6] | ((unsigned)lo[7] << 16);
    bhi[t] = __builtin_bit_cast(kbf8, ph);
    blo[t] = __builtin_bit_cast(kbf8, pl);
  }

  float g[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) g[t][i] = -__builtin_inff();
  float tau[2] = {0.0f, 0.0f};
  int cnt[2] = {0, 0};

  screen_sweep<0, CH, QC, PB>(panel, ctab, nch, abuf, cbuf, q, bhi, blo, g, tau, cnt, tid);
  // tau: the r-th largest of the point's 32 group maxima, less the point's share of the error budget (twice: once for
  // the bounds of pass 1, once for the test of pass 2)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    sort_desc<16>(g[t]);
#pragma unroll
    for (int i = 0; i < 16; ++i) sg[tid * 17 + i] = g[t][i];
    __syncthreads();
    float m = __builtin_inff();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < r) m = fminf(m, fmaxf(g[t][i], sg[(tid ^ 32) * 17 + (r - 1 - i)]));
    }
    tau[t] = off[t] ? __builtin_inff() : m - KNN_SCREEN_MU * xxf[t];
    __syncthreads();
  }
  if (stop_after == 1) { if (tau[0] + tau[1] == 12345.0f) idx_out[0] = 0; return; }
  screen_sweep<1, CH, QC, PB>(panel, ctab + s_pad, nch, abuf, cbuf, q, bhi, blo, g, tau, cnt, tid);
  if (stop_after == 2) { if (cnt[0] + cnt[1] == 123456) idx_out[0] = 0; return; }
#pragma unroll
  for (int t = 0; t < 2; ++t) qcnt[kh * PB + w * 64 + t * 32 + col] = off[t] ? QC + 1 : cnt[t] / (4 * PB);
  __syncthreads();

  // ---- exact: lane = point.  The two halves' queues are ascending in the anchor index: merged on the fly, the strict
  // '<' insertion sees the anchors in the oracle's order.
  long i = base + tid;
  const bool live = i < n;
  if (!live) i = n - 1;
  double x[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (k < d) ? X[(size_t)k * ldx + i] : 0.0;
  double xx = x[0] * x[0];
#pragma unroll
  for (int k = 1; k < DP; ++k) xx = __builtin_fma(x[k], x[k], xx);  // zero padding adds exactly 0
  int c0 = qcnt[tid], c1 = qcnt[PB + tid];
  const bool rescan = c0 > QC || c1 > QC;
  if (rescan) { c0 = 0; c1 = 0; }
  TopList<RCAP> top;
  top.init(r);
  auto row = [&](int j, double (&uk)[DP], double &un) {
    const double *u = Ut + (size_t)j * DP;
#pragma unroll
    for (int k = 0; k < DP; ++k) uk[k] = u[k];
    un = uu[j];
  };
  auto chain = [&](const double (&xa)[DP], double xxa, const double (&uk)[DP], double un) {
    double acc = xa[0] * uk[0];
#pragma unroll
    for (int k = 1; k < DP; ++k) acc = __builtin_fma(xa[k], uk[k], acc);
    return __builtin_fma(-2.0, acc, xxa) + un;
  };
  {
    // entries: (group of eight anchors << 4) | mask of the queue's four (half 0: anchors 8 g .. 8 g + 3, half 1: + 4 .. + 7)
    int ctot = 0;
    for (int e = 0; e < c0; ++e) ctot += __builtin_popcount((unsigned)q[(e * 2 + 0) * PB + tid] & 15u);
    for (int e = 0; e < c1; ++e) ctot += __builtin_popcount((unsigned)q[(e * 2 + 1) * PB + tid] & 15u);
    int p0 = 0, p1 = 0;          // entries taken from either queue
    unsigned m0 = 0, m1 = 0;     // what is left of the entries in hand
    int b0 = 0x7fffffff, b1 = 0x7fffffff;   // their first anchors
    auto refill = [&]() {
      if (!m0 && p0 < c0) { const unsigned e = q[(p0 * 2 + 0) * PB + tid]; m0 = e & 15u; b0 = (int)(e >> 4) * 8; ++p0; }
      if (!m1 && p1 < c1) { const unsigned e = q[(p1 * 2 + 1) * PB + tid]; m1 = e & 15u; b1 = (int)(e >> 4) * 8 + 4; ++p1; }
    };
    auto next = [&]() {          // the next anchor in ascending order: the entry in hand with the smaller first anchor (the groups are disjoint)
      refill();
      const bool first = m0 && (!m1 || b0 < b1);
      unsigned &mm = first ? m0 : m1;
      const int bit = __builtin_ctz(mm | 16u);
      mm &= mm - 1;
      return (first ? b0 : b1) + bit;
    };
    double un[DP], unn = 0.0;
    int jn = 0;
    if (0 < ctot) jn = next();
    row(jn, un, unn);
    for (int e = 0; __any(e < ctot); ++e) {
      double uc[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) uc[k] = un[k];
      const double ucn = unn;
      const int j = jn;
      if (e + 1 < ctot) { jn = next(); row(jn, un, unn); }     // on its way while this one is evaluated
      if (e < ctot) {
        const double D = chain(x, xx, uc, ucn);
        if (D < top.thr()) top.insert(D, j);
      }
    }
  }
  if (live && !rescan) {
#pragma unroll
    for (int k = 0; k < RCAP; ++k) {
      const int slot = k - (RCAP - r);
      if (slot >= 0) {
        const int bj_ = top.bi[k];     // see knn_kernel: never let the sentinel out
        idx_out[(size_t)slot * ldo + i] = ((unsigned)bj_ < (unsigned)s) ? bj_ : slot;
        if (dist_out) dist_out[(size_t)slot * ldo + i] = top.bd[k];
      }
    }
  }

  // ---- points the screen left alone: the wave scans all anchors for one of them at a time
  unsigned long long todo = __ballot(rescan && live);
  while (todo) {
    const int p = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    double xp[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xp[k] = __shfl(x[k], p);
    const double xxp = __shfl(xx, p);
    const long ip = base + w * 64 + p;
    top.init(r);
    for (int j = l; j < s; j += 64) {
      double uk[DP], un;
      row(j, uk, un);
      const double D = chain(xp, xxp, uk, un);
      if (D < top.thr()) top.insert(D, j);
    }
    int head = 0;                       // entries of my list already handed out
    for (int slot = 0; slot < r; ++slot) {
      double v = __builtin_inf();
      int vj = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < RCAP; ++k)
        if (k == RCAP - r + head) { v = top.bd[k]; vj = top.bi[k]; }
      double mv = v;
      int mj = vj;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(mv, o);
        const int oj = __shfl_xor(mj, o);
        if (ov < mv || (ov == mv && oj < mj)) { mv = ov; mj = oj; }
      }
      const bool found = mv < __builtin_inf();
      if (found && v == mv && vj == mj) ++head;
      if (l == 0) {
        idx_out[(size_t)slot * ldo + ip] = found ? mj : slot;
        if (dist_out) dist_out[(size_t)slot * ldo + ip] = mv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k-NN on the matrix cores.  v_mfma_f64_16x16x4_f64 IS the oracle's arithmetic: the instruction
// accumulates its four products as a chain of IEEE FMAs in ascending k, starting from the C
// operand (probed on the device against all 24 orders, 5e6 elements, scripts/probe_mfma_order.hip:
// only k = 0,1,2,3 matches, and it matches bit for bit).  A run of DP/4 instructions over
// k = 0..DP-1 from C = 0 therefore returns exactly  fma(x_{d-1}, u_{d-1}, ... fma(x_1, u_1, x_0 u_0))
// (fma(x_0, u_0, +0) is the rounded product; a zero of the other sign cannot reach a distance), for
// 16 points x 16 anchors at a time, with the operand reuse of a GEMM instead of one broadcast
// operand read per FMA.
//
// One wave owns 64 points: their coordinates stay in VGPRs as four 16-point A fragments; anchors come
// through an LDS tile stored [k][anchor] (the B fragment is then a conflict-free read).  The MFMA
// result layout spreads a point's 16 new dot products over 16 lanes, so they are turned through LDS
// ([point][anchor], row stride 17): lane l then reads the 16 values of ITS point and runs the same
// selection as knn_kernel -- D = fma(-2, dot, |x|^2) + |u|^2 against the r-th best in a register,
// survivors into a private LDS queue (in ascending anchor order, so the strict '<' insertion keeps
// the lower index on ties), sorted top-r list in registers, wave-wide drain when a queue fills.
// ------------------------------------------------------------------------------------------
typedef double kd4 __attribute__((ext_vector_type(4)));

template <int DP, int RCAP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(DP <= 32 ? 2 : 1, 2))) void knn_mfma_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                       const double *__restrict__ Ut,
                                                       const double *__restrict__ uu, int s, int r,
                                                       int *__restrict__ idx_out,
                                                       double *__restrict__ dist_out, int ldo) {
  constexpr int NW = 2;          // waves per workgroup (they share the anchor tile)
  constexpr int TA = 64;         // anchors per LDS tile = 4 groups of 16
  constexpr int TLD = TA + 1;    // [k][anchor] row stride
  constexpr int QC = 8;          // queue slots per point; fullness is checked every four candidates
  constexpr int NQ = DP / 4;     // MFMAs per 16 x 16 block
  constexpr int DLD = 17;        // [point][anchor] row stride of the turned dot products
  __shared__ double tile[DP * TLD];
  __shared__ double dots[NW][64 * DLD];
  __shared__ double q_d[NW][QC * 64];
  __shared__ int q_j[NW][QC * 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const long pbase = (long)blockIdx.x * (64 * NW) + wave * 64;

  // ---- the lane's own point (selection side): |x|^2 by the oracle's chain
  long i_own = pbase + lane;
  const bool live = i_own < n;
  if (!live) i_own = n - 1;     // clamp: computes a duplicate, never stored
  double xx;
  {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double xk = (k < d) ? X[(size_t)k * ldx + i_own] : 0.0;
      acc = (k == 0) ? xk * xk : __builtin_fma(xk, xk, acc);   // zero padding adds exactly 0
    }
    xx = acc;
  }
  TopList<RCAP> top;
  top.init(r);
  int cnt = 0;

  // ---- A fragments: point pt*16 + fr, coordinate 4q + fk
  double xa[4][NQ];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    long i = pbase + pt * 16 + fr;
    if (i >= n) i = n - 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) xa[pt][q] = (4 * q + fk < d) ? X[(size_t)(4 * q + fk) * ldx + i] : 0.0;
  }

  auto drain = [&]() {
    const int c = cnt;
    for (int q = 0; __any(q < c); ++q) {
      if (q < c) {
        const double D = q_d[wave][q * 64 + lane];
        const int j = q_j[wave][q * 64 + lane];
        if (D < top.thr()) top.insert(D, j);
      }
    }
    cnt = 0;
  };

  const int s_pad = (s + 127) / 128 * 128;   // rows of the padded panel (flgp_dev_anchor_rows): zeros, |u|^2 = +inf
  double *mydots = dots[wave];
  for (int j0 = 0; j0 < s_pad; j0 += TA) {
    __syncthreads();
    {
      const double *src = Ut + (size_t)j0 * DP;
      for (int e = tid; e < TA * DP; e += 64 * NW) tile[(e % DP) * TLD + e / DP] = src[e];
    }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < TA / 16; ++g) {
      double ub[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) ub[q] = tile[(4 * q + fk) * TLD + g * 16 + fr];
      // the 16 squared norms of this group: wave-uniform scalar loads, issued ahead of their use
      double uug[16];
#pragma unroll
      for (int a = 0; a < 16; ++a) uug[a] = uu[j0 + g * 16 + a];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        kd4 acc = kd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[pt][q], ub[q], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mydots[(pt * 16 + fk + 4 * reg) * DLD + fr] = acc[reg];
      }
      __threadfence_block();   // one wave: its LDS operations complete in order
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        if ((a & 3) == 0 && __any(cnt > QC - 4)) drain();     // the next four candidates fit for sure
        const double D = __builtin_fma(-2.0, mydots[lane * DLD + a], xx) + uug[a];
        if (D < top.thr()) {
          q_d[wave][cnt * 64 + lane] = D;
          q_j[wave][cnt * 64 + lane] = j0 + g * 16 + a;
          ++cnt;
        }
      }
      __threadfence_block();   // the reads above are done before the next group overwrites the block
    }
  }
  drain();

  if (live) {
#pragma unroll
    for (int k = 0; k < RCAP; ++k) {
      const int slot = k - (RCAP - r);
      if (slot >= 0) {
        const int bj_ = top.bi[k];     // see knn_kernel: never let the sentinel out
        idx_out[(size_t)slot * ldo + i_own] = ((unsigned)bj_ < (unsigned)s) ? bj_ : slot;
        if (dist_out) dist_out[(size_t)slot * ldo + i_own] = top.bd[k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// r = 1 on the matrix cores (Lloyd's assignment step, the cluster counts of subsample_cpp): no list, no queue and
// no turn through LDS.  The MFMA result layout gives lane (fk, fr) the dot products of points fk + 4 reg (reg 0..3)
// of a 16-point block with anchor fr of a 16-anchor group; the lane keeps the running minimum of each of its
// 4 x 4 (block, reg) points over the anchors of ITS residue class (fr mod 16), in ascending anchor order with a
// strict '<' -- the lower index keeps a tie -- and the 16 lanes of a row merge their minima at the very end
// (smaller distance, then smaller index).  Same arithmetic as everywhere: chain from C = 0, D = fma(-2, dot, |x|^2) + |u|^2.
// (Measured why this exists: without any selection the VALU kernel still needs 4.4 ms per 1e6 x 5000 x 16 -- its LDS
// broadcast operand reads, not the selection, are what it waits for.)
// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(128) void knn1_mfma_kernel(const double *__restrict__ X, int n, int ldx, int d,
                                                        const double *__restrict__ Ut, const double *__restrict__ uu, int s,
                                                        int *__restrict__ idx_out, double *__restrict__ dist_out) {
  constexpr int NW = 2, TA = 64, TLD = TA + 1, NQ = DP / 4;
  __shared__ double tile[DP * TLD];
  __shared__ double xxs[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const long pbase = (long)blockIdx.x * (64 * NW) + wave * 64;
  {   // |x|^2 of point pbase + lane by the oracle's chain, handed to the lanes that need it through LDS
    long i = pbase + lane;
    if (i >= n) i = n - 1;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double xk = (k < d) ? X[(size_t)k * ldx + i] : 0.0;
      acc = (k == 0) ? xk * xk : __builtin_fma(xk, xk, acc);
    }
    xxs[wave][lane] = acc;
  }
  double xa[4][NQ];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    long i = pbase + pt * 16 + fr;
    if (i >= n) i = n - 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) xa[pt][q] = (4 * q + fk < d) ? X[(size_t)(4 * q + fk) * ldx + i] : 0.0;
  }
  __syncthreads();
  double xxp[4][4], bd[4][4];
  int bj[4][4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      xxp[pt][reg] = xxs[wave][pt * 16 + fk + 4 * reg];
      bd[pt][reg] = __builtin_inf();
      bj[pt][reg] = 0x7fffffff;
    }
  const int s_pad = (s + 127) / 128 * 128;   // padded panel: zeros, |u|^2 = +inf (never below a minimum)
  for (int j0 = 0; j0 < s_pad; j0 += TA) {
    __syncthreads();
    {
      const double *src = Ut + (size_t)j0 * DP;
      for (int e = tid; e < TA * DP; e += 64 * NW) tile[(e % DP) * TLD + e / DP] = src[e];
    }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < TA / 16; ++g) {
      double ub[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) ub[q] = tile[(4 * q + fk) * TLD + g * 16 + fr];
      const int j = j0 + g * 16 + fr;
      const double uuj = uu[j];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        kd4 acc = kd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[pt][q], ub[q], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const double D = __builtin_fma(-2.0, acc[reg], xxp[pt][reg]) + uuj;
          const bool c = D < bd[pt][reg];
          bd[pt][reg] = c ? D : bd[pt][reg];
          bj[pt][reg] = c ? j : bj[pt][reg];
        }
      }
    }
  }
  // merge the 16 residue classes of every point: lanes fk*16 .. fk*16+15
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      double dmin = bd[pt][reg];
      int jmin = bj[pt][reg];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        const double od = __shfl_xor(dmin, m, 64);
        const int oj = __shfl_xor(jmin, m, 64);
        const bool take = (od < dmin) || (od == dmin && oj < jmin);
        dmin = take ? od : dmin;
        jmin = take ? oj : jmin;
      }
      const long i = pbase + pt * 16 + fk + 4 * reg;
      if (fr == 0 && i < n) {
        idx_out[i] = ((unsigned)jmin < (unsigned)s) ? jmin : 0;   // (NaN row: see knn_kernel)
        if (dist_out) dist_out[i] = dmin;
      }
    }
}

template <int DP>
static int launch_knn1_mfma(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt,
                            const double *duu, int s, int *d_idx, double *d_dist) {
  ProfScope ps("knn_kernel", st, 2.0 * (double)n * (double)s * (double)d);
  hipLaunchKernelGGL((knn1_mfma_kernel<DP>), dim3(ceil_div(n, 128)), dim3(128), 0, st, dX, n, ldx, d, dUt, duu, s, d_idx,
                     d_dist);
  return check_launch("knn1_mfma_kernel");
}

template <int DP, int RCAP>
static int launch_knn_mfma(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt,
                           const double *duu, int s, int r, int *d_idx, double *d_dist, int ldo) {
  ProfScope ps("knn_kernel", st, 2.0 * (double)n * (double)s * (double)d);
  hipLaunchKernelGGL((knn_mfma_kernel<DP, RCAP>), dim3(ceil_div(n, 128)), dim3(128), 0, st, dX, n, ldx, d, dUt, duu,
                     s, r, d_idx, d_dist, ldo);
  return check_launch("knn_mfma_kernel");
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

template <int DP, int RCAP>
static int launch_knn_screen(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt,
                             const double *duu, int s, int r, int *d_idx, double *d_dist, int ldo) {
  ProfScope ps("knn_kernel", st, 2.0 * (double)n * (double)s * (double)d);
  hipLaunchKernelGGL((knn_screen_kernel<DP, RCAP>), dim3(ceil_div(n, 256)), dim3(256), 0, st, dX, n, ldx, d, dUt, duu, s,
                     r, d_idx, d_dist, ldo, tuning("knn_screen_stop", 0));
  return check_launch("knn_screen_kernel");
}

}  // namespace flgp

namespace flgp {
int knn_wide(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad, const double *duu, int s,
             int r, int *d_idx, double *d_dist, int ldo);   // knn_wide.hip
}

using namespace flgp;

extern "C" int flgp_dev_anchor_dpad(int d) {
  if (d <= 4) return 4;
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  if (d <= 32) return 32;
  if (d <= 64) return 64;
  if (d <= FLGP_DMAX) return (d + 7) / 8 * 8;   // rows of whole 64-byte lines: the GEMM route of knn_wide.hip and the LDS-free LAE
  return -1;
}

// rows to allocate for the anchor panel (Ut: rows x dpad doubles, uu: rows doubles): s rounded up to the kernels' anchor
// tile, and twice as much again for the screening copy behind it (anchor_prep_kernel)
extern "C" int flgp_dev_anchor_rows(int s) { return 3 * anchor_pad_rows(s); }

extern "C" int flgp_dev_anchor_prep(void *stream, const double *dU, int s, int ldu, int d, double *dUt,
                                    double *duu) {
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0 && d >= 1, "k-NN kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  FLGP_REQUIRE(s >= 1 && ldu >= s, "anchor_prep: need s >= 1 and ldu >= s");
  const int s_pad = anchor_pad_rows(s);
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
  if (dpad > 64) return knn_wide(st, dX, n, ldx, d, dUt, dpad, duu, s, r, d_idx, d_dist, ldo);
  const int rcap = r <= 4 ? 4 : (r <= 8 ? 8 : (r <= 16 ? 16 : 32));
  const int variant = tuning("knn_variant", 0);
  if (r == 1 && tuning("knn_nn1", 1)) {   // the 1-NN of Lloyd's assignment step and of the cluster counts
    const int v1 = tuning("knn_nn1_variant", 0);
    if (v1 == 0 || v1 == 5) {
      if (dpad == 4) return launch_knn1_mfma<4>(st, dX, n, ldx, d, dUt, duu, s, d_idx, d_dist);
      if (dpad == 8) return launch_knn1_mfma<8>(st, dX, n, ldx, d, dUt, duu, s, d_idx, d_dist);
      if (dpad == 16) return launch_knn1_mfma<16>(st, dX, n, ldx, d, dUt, duu, s, d_idx, d_dist);
      if (dpad == 32) return launch_knn1_mfma<32>(st, dX, n, ldx, d, dUt, duu, s, d_idx, d_dist);
      if (dpad == 64) return launch_knn1_mfma<64>(st, dX, n, ldx, d, dUt, duu, s, d_idx, d_dist);
    }
    if (dpad == 4) return launch_knn<4, 1, 2, 4, 2, 12>(KNN_ARGS);
    if (dpad == 8) return launch_knn<8, 1, 2, 4, 4, 12>(KNN_ARGS);
    if (dpad == 16) {
      if (v1 == 1) return launch_knn<16, 1, 1, 2, 8, 8>(KNN_ARGS);
      if (v1 == 2) return launch_knn<16, 1, 1, 4, 8, 8>(KNN_ARGS);
      if (v1 == 3) return launch_knn<16, 1, 2, 4, 8, 8>(KNN_ARGS);
      if (v1 == 4) return launch_knn<16, 1, 4, 2, 8, 8>(KNN_ARGS);
      return launch_knn<16, 1, 4, 2, 8, 8>(KNN_ARGS);
    }
    if (dpad == 32) return launch_knn<32, 1, 1, 2, 16, 8>(KNN_ARGS);
    if (dpad == 64) return launch_knn<64, 1, 1, 2, 16, 8>(KNN_ARGS);
  }
  // The matrix-core kernel wins once the distance itself dominates (measured, n = 4e5, s = 5000, r = 10:
  // d = 64 11.4 vs 18.8 ms, d = 32 5.4 vs 5.8 ms); at d <= 16 the selection is the larger half of either
  // kernel and the VALU kernel's three waves per SIMD hide its latencies better (d = 16: 6.7 vs 6.0 ms
  // per 1e6 points).  knn_mfma = 1 / 0 forces one or the other.
  // d <= 16, r <= 16, enough anchors for 32 groups of them: the matrix-core screen (knn_screen = 0 switches it off)
  if (dpad <= 16 && r >= 2 && r <= 16 && s >= 512 && s <= 32768 && tuning("knn_screen", 1) &&
      tuning("knn_mfma", -1) < 0 && variant == 0) {
#define KNN_SCREEN_CASE(DPv, RCv) if (dpad == DPv && rcap == RCv) return launch_knn_screen<DPv, RCv>(KNN_ARGS);
    KNN_SCREEN_CASE(4, 4) KNN_SCREEN_CASE(4, 8) KNN_SCREEN_CASE(4, 16)
    KNN_SCREEN_CASE(8, 4) KNN_SCREEN_CASE(8, 8) KNN_SCREEN_CASE(8, 16)
    KNN_SCREEN_CASE(16, 4) KNN_SCREEN_CASE(16, 8) KNN_SCREEN_CASE(16, 16)
#undef KNN_SCREEN_CASE
  }
  const int use_mfma = tuning("knn_mfma", -1);
  if (use_mfma == 1 || (use_mfma < 0 && dpad >= 32)) {
#define KNN_MFMA_CASE(DPv, RCv) if (dpad == DPv && rcap == RCv) return launch_knn_mfma<DPv, RCv>(KNN_ARGS);
    KNN_MFMA_CASE(4, 4) KNN_MFMA_CASE(4, 8) KNN_MFMA_CASE(4, 16) KNN_MFMA_CASE(4, 32)
    KNN_MFMA_CASE(8, 4) KNN_MFMA_CASE(8, 8) KNN_MFMA_CASE(8, 16) KNN_MFMA_CASE(8, 32)
    KNN_MFMA_CASE(16, 4) KNN_MFMA_CASE(16, 8) KNN_MFMA_CASE(16, 16) KNN_MFMA_CASE(16, 32)
    KNN_MFMA_CASE(32, 4) KNN_MFMA_CASE(32, 8) KNN_MFMA_CASE(32, 16) KNN_MFMA_CASE(32, 32)
    KNN_MFMA_CASE(64, 4) KNN_MFMA_CASE(64, 8) KNN_MFMA_CASE(64, 16) KNN_MFMA_CASE(64, 32)
  }
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
