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
