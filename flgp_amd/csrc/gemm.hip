// k7: fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64) and the heat-kernel contraction built on it.
//
//   flgp_dev_gemm : C(i,j) = alpha * sum_k A(i,k) B(k,j) + beta * E(i,j), arbitrary element
//                   strides, optional split-K with a fixed-order reduction (deterministic).
//   flgp_dev_hk   : H = V[idx0,0:K] diag(exp(-t(1-values))) V[idx1,0:K]^T
//                   (HK_from_spectrum_cpp, reference src/Spectrum.cpp:83-94): the diagonal is
//                   folded into the small operand once, then one GEMM streams V once and
//                   writes H once.
//
// Tiling (gfx950): 128 x 128 block tile, BK = 16, 256 threads = 4 waves in 2 x 2, each wave a
// 64 x 64 sub-tile = 4 x 4 MFMA tiles (16 accumulators x 4 f64 = 128 VGPRs).  Operand tiles go
// global -> registers (prefetched one tile ahead, so HBM latency hides under the MFMAs) -> LDS
// as [k][row] with the row stride padded by 16 doubles (k and k+1 then sit 32 banks apart and
// the ds_read_b64 fragment reads are conflict free).  The fp64 MFMA is slow enough per
// instruction (2048 flop per wave per ~64 cycles) that 8 fragment reads per 16 MFMAs keep the
// matrix pipe fed; two blocks per CU cover the barrier bubbles.
//
// Layout trick: the MFMA result puts the tile COLUMN on lane&15, so for coalesced stores the
// host wrapper orients the problem (computing C^T = B^T A^T when C is column-major) such that
// the memory-contiguous output dimension is always the kernel's column dimension.
#include "common.h"
#include <type_traits>

namespace flgp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int GB = 128;        // block tile (rows and cols) of the large shapes; 64 where 128 leaves CUs idle (GBT below)
constexpr int GK = 16;         // k depth per stage
#ifndef GEMM_GLD_PAD
#define GEMM_GLD_PAD 17
#endif
constexpr int gld_of(int gbt) { return gbt + GEMM_GLD_PAD; }   // padded LDS row stride (doubles)

struct GemmArgs {
  int M, N, Kd;
  const double *A; long a_is, a_ks;   // A(i,k)
  const double *B; long b_ks, b_js;   // B(k,j)
  double alpha, beta, gamma;
  const double *E; long e_is, e_js;    // + beta * E(i,j)
  const double *E2;                    // + gamma * E2(i,j), same strides as E
  double *C; long c_is, c_js;
  int shift_edges;                    // edge tiles slide back inside the matrix (they recompute a few columns / rows)
  int ksplit_len;                     // k range per blockIdx.z (multiple of GK); partials when gridDim.z > 1
  double *part;                       // [z][M][N] row-major partials
  // in-kernel split-K reduction (optional): one counter per output tile, all zero between launches.  The pieces of a
  // tile store their accumulators as planes of PLANE doubles in the thread order they are held in, take a ticket, and
  // the piece that arrives last adds the planes in ascending z (a fixed order: deterministic) and runs the epilogue.
  int *tickets;
  // a second product of the same shape in the same launch (gridDim.y = 2): its own operands and result, everything else shared
  const double *A2, *B2;
  double *C2;
};

constexpr int PLANE = GB * GB;    // doubles per partial plane of one 128 x 128 tile

// 16-byte store that is written through to the memory side (sc1), so that a workgroup on another XCD can read it
// after the ticket hand-off without a release fence over this XCD's whole L2 (MI355X guide: "publish-large")
__device__ __forceinline__ void store_wt_d2(double *p, d2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

// ---- operand staging: a 128(rows) x 16(k) tile goes global -> 8 registers per thread -> LDS [k][row].
// Three thread mappings, chosen per operand and per stage (uniform over the workgroup):
//   RC  row-contiguous operand, tile and stage fully inside: each thread loads two adjacent rows at once
//       (one 16-byte load; k = tid>>6 + 4 rep) -- four loads and four LDS stores per stage;
//   KC  k-contiguous operand, fully inside: two adjacent k of one row per load (row = tid>>3 + 32 rep);
//   GEN anything else (edge tiles, the k tail, arbitrary strides): one element per load, out-of-range
//       elements read a clamped address and are zeroed by a select.
// Measured on the 5120 x 256 x 5120 product: the predicated one-element path costs 14 % of the run time,
// which is why the interior of the problem does not take it.
enum { LOAD_GEN = 0, LOAD_RC = 1, LOAD_KC = 2 };

template <int GBT>
__device__ __forceinline__ void tile_load_gen(double (&reg)[GBT / 16], const double *__restrict__ base, long rs, long ks,
                                              int row0, int nrows, int k0, int kend, bool row_contig, int tid) {
  constexpr int NR = GBT / 16, KPT = 256 / GBT;   // registers per thread; k values covered by the 256 threads at once
  if (row_contig) {
    const int row = row0 + (tid & (GBT - 1));
    const int kb = tid / GBT;
    const bool rv = row < nrows;
    const double *pr = base + (size_t)(rv ? row : nrows - 1) * rs;
#pragma unroll
    for (int rep = 0; rep < NR; ++rep) {
      const int k = k0 + kb + KPT * rep;
      const bool kv = k < kend;
      const double v = pr[(size_t)(kv ? k : kend - 1) * ks];
      reg[rep] = (rv && kv) ? v : 0.0;
    }
  } else {
    const int k = k0 + (tid & 15);
    const int rb = tid >> 4;
    const bool kv = k < kend;
    const double *pk = base + (size_t)(kv ? k : kend - 1) * ks;
#pragma unroll
    for (int rep = 0; rep < NR; ++rep) {
      const int row = row0 + rb + 16 * rep;
      const bool rv = row < nrows;
      const double v = pk[(size_t)(rv ? row : nrows - 1) * rs];
      reg[rep] = (rv && kv) ? v : 0.0;
    }
  }
}

template <int GBT>
__device__ __forceinline__ void tile_store_gen(const double (&reg)[GBT / 16], double *__restrict__ lds, bool row_contig, int tid) {
  constexpr int NR = GBT / 16, KPT = 256 / GBT, GLD = gld_of(GBT);
  if (row_contig) {
    const int row = tid & (GBT - 1), kb = tid / GBT;
#pragma unroll
    for (int rep = 0; rep < NR; ++rep) lds[(kb + KPT * rep) * GLD + row] = reg[rep];
  } else {
    const int k = tid & 15, rb = tid >> 4;
#pragma unroll
    for (int rep = 0; rep < NR; ++rep) lds[k * GLD + rb + 16 * rep] = reg[rep];
  }
}

struct Operand {
  const double *base;   // element (row, k) at base[row * rs + k * ks]
  long rs, ks;
  int row0, nrows;
  int mode;             // LOAD_RC / LOAD_KC when the whole tile is inside and the layout allows, else LOAD_GEN
  bool row_contig;
  const double *fast;   // per-thread pointer of the fast mappings at k = 0
};

// fast mappings, GBT rows x 16 k over 256 threads:
//   RC: a thread takes rows 2 l, 2 l + 1 (l = tid % (GBT/2)) at k = tid / (GBT/2) + (512/GBT) rep, rep < GBT/32
//   KC: a thread takes k = 2 (tid & 7), + 1 of row tid >> 3 + 32 rep, rep < GBT/32
template <int GBT>
__device__ __forceinline__ Operand make_operand(const double *base, long rs, long ks, int row0, int nrows, int tid) {
  Operand o;
  o.base = base; o.rs = rs; o.ks = ks; o.row0 = row0; o.nrows = nrows;
  o.row_contig = (rs == 1);
  o.mode = LOAD_GEN;
  o.fast = base;
  const bool inside = row0 + GBT <= nrows;
  const bool al16 = (((size_t)base) & 15) == 0;
  if (inside && al16 && rs == 1 && (ks & 1) == 0) {
    o.mode = LOAD_RC;
    o.fast = base + (size_t)(row0 + 2 * (tid & (GBT / 2 - 1))) + (size_t)(tid / (GBT / 2)) * ks;
  } else if (inside && al16 && ks == 1 && (rs & 1) == 0) {
    o.mode = LOAD_KC;
    o.fast = base + (size_t)(row0 + (tid >> 3)) * rs + (size_t)(2 * (tid & 7));
  }
  return o;
}

// stage [k0, k0 + GK) of the operand into registers; `full` = the stage lies inside [., kend)
template <int GBT>
__device__ __forceinline__ void stage_load(double (&reg)[GBT / 16], const Operand &o, int k0, int kend, bool full, int tid) {
  constexpr int NP = GBT / 32, KS = 512 / GBT;     // 16-byte loads per thread; k step between them in the RC mapping
  if (full && o.mode == LOAD_RC) {
    const double *p = o.fast + (size_t)k0 * o.ks;
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const d2 v = *(const d2 *)(p + (size_t)(KS * rep) * o.ks);
      reg[2 * rep] = v[0];
      reg[2 * rep + 1] = v[1];
    }
  } else if (full && o.mode == LOAD_KC) {
    const double *p = o.fast + k0;
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const d2 v = *(const d2 *)(p + (size_t)(32 * rep) * o.rs);
      reg[2 * rep] = v[0];
      reg[2 * rep + 1] = v[1];
    }
  } else if (o.mode == LOAD_RC) {
    // the k tail of a row-contiguous operand keeps the wide mapping: k = k0 + tid / (GBT/2) + KS rep is the same for
    // GBT/2 adjacent lanes, rows past kend read row kend - 1 (a valid address) and are zeroed
    const int kt = tid / (GBT / 2);
    const int kq = k0 + kt;
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const int k = kq + KS * rep;
      const int kc = (k < kend) ? k : kend - 1;
      const d2 v = *(const d2 *)(o.fast + (size_t)(kc - kt) * o.ks);   // o.fast sits at k = kt
      reg[2 * rep] = (k < kend) ? v[0] : 0.0;
      reg[2 * rep + 1] = (k < kend) ? v[1] : 0.0;
    }
  } else {
    tile_load_gen<GBT>(reg, o.base, o.rs, o.ks, o.row0, o.nrows, k0, kend, o.row_contig, tid);
  }
}

template <int GBT>
__device__ __forceinline__ void stage_store(const double (&reg)[GBT / 16], double *__restrict__ lds, const Operand &o, bool full,
                                            int tid) {
  constexpr int NP = GBT / 32, KS = 512 / GBT, GLD = gld_of(GBT);
  if (o.mode == LOAD_RC) {      // (the tail stage of a row-contiguous operand uses this mapping too)
    double *q = lds + (tid / (GBT / 2)) * GLD + 2 * (tid & (GBT / 2 - 1));
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      q[(KS * rep) * GLD] = reg[2 * rep];
      q[(KS * rep) * GLD + 1] = reg[2 * rep + 1];
    }
  } else if (full && o.mode == LOAD_KC) {
    double *q = lds + (2 * (tid & 7)) * GLD + (tid >> 3);
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      q[32 * rep] = reg[2 * rep];
      q[GLD + 32 * rep] = reg[2 * rep + 1];
    }
  } else {
    tile_store_gen<GBT>(reg, lds, o.row_contig, tid);
  }
}

// The fast mappings again, for a pipeline that keeps several stages in flight: every load is issued unconditionally (a
// stage past the end of the k range reads clamped addresses) and the zeroing of the padding happens when the registers go to
// LDS.  stage_load above chooses its mapping and its tail handling at run time; behind those branches the compiler's
// wait-count pass cannot tell how many loads are outstanding and waits for all of them (vmcnt(0)) before every LDS store,
// which turns a ring of register sets back into one stage of flight (measured on the 64-tile: 1.5 us per stage with one
// set or with three).  kend must be even in the KC mapping (a thread loads two adjacent k).
template <int GBT, int MODE>
__device__ __forceinline__ void fast_load(double (&reg)[GBT / 16], const Operand &o, int k0, int kend, int tid) {
  constexpr int NP = GBT / 32, KS = 512 / GBT;
  if constexpr (MODE == LOAD_RC) {
    const int kt = tid / (GBT / 2);
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const int k = k0 + kt + KS * rep;
      const int kc = (k < kend) ? k : kend - 1;
      const d2 v = *(const d2 *)(o.fast + (size_t)(kc - kt) * o.ks);   // o.fast sits at k = kt
      reg[2 * rep] = v[0];
      reg[2 * rep + 1] = v[1];
    }
  } else {
    const int kl = 2 * (tid & 7);
    const int k = k0 + kl;
    const int kc = (k < kend) ? k : kend - 2;
    const double *p = o.fast + (kc - kl);                               // o.fast sits at k = kl
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const d2 v = *(const d2 *)(p + (size_t)(32 * rep) * o.rs);
      reg[2 * rep] = v[0];
      reg[2 * rep + 1] = v[1];
    }
  }
}
template <int GBT, int MODE>
__device__ __forceinline__ void fast_store(const double (&reg)[GBT / 16], double *__restrict__ lds, int k0, int kend, int tid) {
  constexpr int NP = GBT / 32, KS = 512 / GBT, GLD = gld_of(GBT);
  if constexpr (MODE == LOAD_RC) {
    const int kt = tid / (GBT / 2);
    double *q = lds + kt * GLD + 2 * (tid & (GBT / 2 - 1));
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      const bool in = k0 + kt + KS * rep < kend;
      q[(KS * rep) * GLD] = in ? reg[2 * rep] : 0.0;
      q[(KS * rep) * GLD + 1] = in ? reg[2 * rep + 1] : 0.0;
    }
  } else {
    const bool in = k0 + 2 * (tid & 7) < kend;
    double *q = lds + (2 * (tid & 7)) * GLD + (tid >> 3);
#pragma unroll
    for (int rep = 0; rep < NP; ++rep) {
      q[32 * rep] = in ? reg[2 * rep] : 0.0;
      q[GLD + 32 * rep] = in ? reg[2 * rep + 1] : 0.0;
    }
  }
}

// The inner product runs on v_mfma_f64_16x16x4_f64.  (The 4x4x4 four-block form was tried as well: both
// forms sustain 72-76 TFLOP/s in the bare inner loop -- scripts/ubench_inner.hip -- so the form with the
// fewer operand reads stays.  Layout of the 4x4x4 form, probed on the device with scripts/probe_mfma4.hip:
// A_b(i,k) in lane i + 4b + 16k, B_b(k,j) in lane j + 4b + 16k, D_b(i,j) in lane j + 4b + 16i.)
// GBT = 128: the tile of the large shapes.  GBT = 64 (each wave a 32 x 32 sub-tile, 2 x 2 MFMA tiles): the solver's
// s x b and b x b products, which have 80 and 4 tiles of 128 -- a quarter of the chip, or split-K planes and a reduction
// launch to make up for it -- and 316 / 16 tiles of 64.
template <int GBT>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  constexpr int GB = GBT, GLD = gld_of(GBT), MI = GBT / 32, NR = GBT / 16;
  // two LDS stages: the tile of stage s+1 is written while stage s is being multiplied, one barrier per stage
  __shared__ double As2[2][GK * GLD];
  __shared__ double Bs2[2][GK * GLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * (GBT / 2), wc = (wave & 1) * (GBT / 2);

  // XCD-aware tile order: consecutive block ids share an XCD every 8; give each XCD a band of tiles
  const int ntm = (g.M + GB - 1) / GB, ntn = (g.N + GB - 1) / GB;
  const int nt = ntm * ntn;
  int bid = blockIdx.x;
  const double *gA = g.A, *gB = g.B;
  double *gC = g.C;
  if (gridDim.y > 1) {   // two products: the bands run over both tile sets (workgroups go to the XCDs by their linear id)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, nt2 = 2 * nt;
    const int q = nt2 / 8, rem = nt2 % 8, xcd = lin % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + lin / 8;
    if (bid >= nt) { bid -= nt; gA = g.A2; gB = g.B2; gC = g.C2; }
  } else {
    const int q = nt / 8, rem = nt % 8, xcd = bid % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
  }
  // the shorter tile dimension runs fastest, so the blocks that are resident together on an XCD
  // share the long operand's panel through L2 and the short operand stays L2 resident
  int tm = (ntm <= ntn) ? bid % ntm : bid / ntn;
  int tn = (ntm <= ntn) ? bid / ntm : bid % ntn;
  const int zidx = blockIdx.z;
  int row0 = tm * GB, col0 = tn * GB;
  // An edge tile would stage its operands through the predicated path for all of its k range and, being
  // the slowest block, set the run time (measured: 286 us at 4992, 351 us at 5000).  Where it is safe the
  // tile is moved back inside instead and recomputes rows / columns its neighbour also writes -- the same
  // values, from the same operations in the same order.
  if (g.shift_edges) {
    if (row0 + GB > g.M && g.M >= GB) row0 = g.M - GB;
    if (col0 + GB > g.N && g.N >= GB) col0 = g.N - GB;
  }

  // the k stages of this block: a contiguous range
  int kbeg = blockIdx.z * g.ksplit_len, kend = kbeg + g.ksplit_len;
  if (kend > g.Kd) kend = g.Kd;
  int ns = (kend - kbeg + GK - 1) / GK;
  if (ns < 0) ns = 0;
  auto kof = [&](int si) { return kbeg + si * GK; };

  const Operand oa = make_operand<GBT>(gA, g.a_is, g.a_ks, row0, g.M, tid);
  const Operand ob = make_operand<GBT>(gB, g.b_js, g.b_ks, col0, g.N, tid);
  d4 acc[MI][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < MI; ++ni) acc[mi][ni] = d4{0.0, 0.0, 0.0, 0.0};

  // Operand stages on their way: NS register sets.  While stage si is multiplied out of one LDS buffer, set (si + 1) % NS
  // (stage si + 1, loaded NS stages ago) goes into the other buffer and is refilled with stage si + 1 + NS.  The 128-tile
  // has registers for one set (two stages of flight: its 64 MFMAs per stage cover the latency); the 64-tile does 16 MFMAs
  // per stage and, on the shapes it is used for, runs alone on its CU: three sets, four stages of flight -- with the
  // unconditional loads of fast_load (see there), else the sets buy nothing.
  constexpr int NS = (GBT == 64) ? 3 : 1;
  const int fr = lane & 15, fk = lane >> 4;
  auto pipeline = [&](auto fast_c, auto ma_c, auto mb_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    constexpr int MA = decltype(ma_c)::value, MB = decltype(mb_c)::value;
    double ra[NS][NR], rb[NS][NR];
    auto load = [&](int set, int si) {          // stage si (clamped to the last one when FAST: the loads stay unconditional)
      if constexpr (FAST) {
        const int k0 = kof(si < ns ? si : ns - 1);
        fast_load<GBT, MA>(ra[set], oa, k0, kend, tid);
        fast_load<GBT, MB>(rb[set], ob, k0, kend, tid);
      } else if (si < ns) {
        const int k0 = kof(si);
        const bool f = k0 + GK <= kend;
        stage_load<GBT>(ra[set], oa, k0, kend, f, tid);
        stage_load<GBT>(rb[set], ob, k0, kend, f, tid);
      }
    };
    auto store = [&](int set, int si, int buf) {
      const int k0 = kof(si);
      if constexpr (FAST) {
        fast_store<GBT, MA>(ra[set], As2[buf], k0, kend, tid);
        fast_store<GBT, MB>(rb[set], Bs2[buf], k0, kend, tid);
      } else {
        const bool f = k0 + GK <= kend;
        stage_store<GBT>(ra[set], As2[buf], oa, f, tid);
        stage_store<GBT>(rb[set], Bs2[buf], ob, f, tid);
      }
    };
    if (ns > 0) {
      load(0, 0);
      store(0, 0, 0);
#pragma unroll
      for (int j = 1; j <= NS; ++j) load(j % NS, j);
    }
    __syncthreads();
    int cur = 0;
    auto step = [&](auto set_c, int si) {
      constexpr int SET = decltype(set_c)::value;     // == (si + 1) % NS
      const double *As = As2[cur], *Bs = Bs2[cur];
      // stage si+1 goes into the other buffer (last read before the previous barrier) and stage si+1+NS starts its way
      // from HBM / L2.  In the FAST pipeline both happen on every step, past the end too (a repeat of the last stage into a
      // buffer nobody reads any more): with the loads under a condition the wait-count pass gives up on counting them.
      if (FAST || si + 1 < ns) {
        store(SET, (FAST && si + 1 >= ns) ? ns - 1 : si + 1, cur ^ 1);
        load(SET, si + 1 + NS);
      }
      // (the all-padding k steps of a ragged last stage are multiplied like the others: for K = 200 skipping them removed
      //  4 % of the MFMAs and not a microsecond, and it puts a branch between the k steps)
      if (si < ns) {
#pragma unroll
      for (int kk = 0; kk < GK; kk += 4) {
        double fa[MI], fb[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = As[(kk + fk) * GLD + wr + mi * 16 + fr];
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) fb[ni] = Bs[(kk + fk) * GLD + wc + ni * 16 + fr];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < MI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      }
      }
      __syncthreads();
      cur ^= 1;
    };
    if constexpr (NS == 1) {
      for (int si = 0; si < ns; ++si) step(std::integral_constant<int, 0>{}, si);
    } else {
      static_assert(NS == 3, "the loop below is unrolled for three register sets");
      for (int si = 0; si < ns; si += 3) {
        step(std::integral_constant<int, 1>{}, si);
        if (FAST || si + 1 < ns) step(std::integral_constant<int, 2>{}, si + 1);
        if (FAST || si + 2 < ns) step(std::integral_constant<int, 0>{}, si + 2);
      }
    }
  };
  using std::integral_constant;
  typedef integral_constant<bool, true> yes_t;
  typedef integral_constant<bool, false> no_t;
  typedef integral_constant<int, LOAD_RC> rc_t;
  typedef integral_constant<int, LOAD_KC> kc_t;
  bool done = false;
  if constexpr (GBT == 64) {
    if (ns > 0 && (kend & 1) == 0 && kend - kbeg >= 2 && oa.mode != LOAD_GEN && ob.mode != LOAD_GEN) {
      if (oa.mode == LOAD_RC && ob.mode == LOAD_RC) pipeline(yes_t{}, rc_t{}, rc_t{});
      else if (oa.mode == LOAD_RC) pipeline(yes_t{}, rc_t{}, kc_t{});
      else if (ob.mode == LOAD_RC) pipeline(yes_t{}, kc_t{}, rc_t{});
      else pipeline(yes_t{}, kc_t{}, kc_t{});
      done = true;
    }
  }
  if (!done) pipeline(no_t{}, rc_t{}, rc_t{});

  // epilogue: D(row = (lane>>4) + 4*reg, col = lane&15) of each 16x16 tile
  const bool partial = gridDim.z > 1;
  if constexpr (GBT == 128) if (partial && g.tickets) {
    // ---- split-K finished inside the kernel
    __shared__ int last_piece;
    const int tile_id = tm * ntn + tn;
    const int npieces = (int)gridDim.z;
    if (npieces > 1) {
      double *plane = g.part + ((size_t)zidx * nt + tile_id) * PLANE;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            store_wt_d2(plane + ((size_t)((mi * 4 + ni) * 2 + h) * 256 + tid) * 2, d2{acc[mi][ni][2 * h], acc[mi][ni][2 * h + 1]});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&g.tickets[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_piece = (old == npieces - 1);
        if (old == npieces - 1) {
          __hip_atomic_store(&g.tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();
      if (!last_piece) return;
      const double *p0 = g.part + (size_t)tile_id * PLANE + (size_t)tid * 2;
      const size_t zstride = (size_t)nt * PLANE;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        d2 sum[4][2];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int h = 0; h < 2; ++h) sum[ni][h] = d2{0.0, 0.0};
        for (int z = 0; z < npieces; ++z) {   // ascending z: the order does not depend on which piece came last
          d2 v[4][2];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              v[ni][h] = __builtin_nontemporal_load((const d2 *)(p0 + z * zstride + (size_t)((mi * 4 + ni) * 2 + h) * 512));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) sum[ni][h] += v[ni][h];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          acc[mi][ni][0] = sum[ni][0][0]; acc[mi][ni][1] = sum[ni][0][1];
          acc[mi][ni][2] = sum[ni][1][0]; acc[mi][ni][3] = sum[ni][1][1];
        }
      }
    }
  }
  // The 16 MI^2 results of a thread go out through one base pointer and compile-time multiples of the two strides; which
  // terms the epilogue has (planes / E / E2) and whether the tile needs bounds checks is decided once, not per element
  // (per element it was a chain of scalar branches and three 64-bit multiply-adds: ~20 instructions each, 3 us per tile).
  const bool to_planes = partial && !g.tickets;
  const int ti0 = row0 + wr + fk, tj0 = col0 + wc + fr;
  auto emit = [&](auto pl_c, auto he_c, auto he2_c, auto in_c) {
    constexpr bool PL = decltype(pl_c)::value, HE = decltype(he_c)::value, HE2 = decltype(he2_c)::value, IN = decltype(in_c)::value;
    const long cis = PL ? (long)g.N : g.c_is, cjs = PL ? 1L : g.c_js;
    double *cb = (PL ? g.part + (size_t)zidx * g.M * g.N : gC) + (size_t)ti0 * cis + (size_t)tj0 * cjs;
    const double *eb = HE ? g.E + (size_t)ti0 * g.e_is + (size_t)tj0 * g.e_js : nullptr;
    const double *e2b = HE2 ? g.E2 + (size_t)ti0 * g.e_is + (size_t)tj0 * g.e_js : nullptr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int ni = 0; ni < MI; ++ni) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int di = mi * 16 + 4 * reg, dj = ni * 16;
          if (!IN && (ti0 + di >= g.M || tj0 + dj >= g.N)) continue;
          const double v = acc[mi][ni][reg];
          if constexpr (PL) {
            cb[(size_t)di * cis + (size_t)dj * cjs] = v;
          } else {
            double o = g.alpha * v;
            if constexpr (HE) o += g.beta * eb[(size_t)di * g.e_is + (size_t)dj * g.e_js];
            if constexpr (HE2) o += g.gamma * e2b[(size_t)di * g.e_is + (size_t)dj * g.e_js];
            cb[(size_t)di * cis + (size_t)dj * cjs] = o;
          }
        }
      }
    }
  };
  {
    typedef std::integral_constant<bool, true> T_;
    typedef std::integral_constant<bool, false> F_;
    const bool inside = row0 + GB <= g.M && col0 + GB <= g.N;
    auto pick_in = [&](auto pl_c, auto he_c, auto he2_c) { if (inside) emit(pl_c, he_c, he2_c, T_{}); else emit(pl_c, he_c, he2_c, F_{}); };
    if (to_planes) pick_in(T_{}, F_{}, F_{});
    else if (g.E && g.E2) pick_in(F_{}, T_{}, T_{});
    else if (g.E) pick_in(F_{}, T_{}, F_{});
    else if (g.E2) pick_in(F_{}, F_{}, T_{});
    else pick_in(F_{}, F_{}, F_{});
  }
}

__global__ void splitk_reduce_kernel(GemmArgs g, int nsplit) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)g.M * g.N) return;
  const int i = (int)(idx / g.N), j = (int)(idx % g.N);
  double v = 0.0;
  for (int z = 0; z < nsplit; ++z) v += g.part[((size_t)z * g.M + i) * g.N + j];  // fixed order
  double o = g.alpha * v;
  if (g.E) o += g.beta * g.E[(size_t)i * g.e_is + (size_t)j * g.e_js];
  if (g.E2) o += g.gamma * g.E2[(size_t)i * g.e_is + (size_t)j * g.e_js];
  g.C[(size_t)i * g.c_is + (size_t)j * g.c_js] = o;
}

// The reduction of a split SQUARE product with the small-matrix work of the eigensolver folded in (GemmFusedReduce,
// common.h).  One workgroup per 16 x 16 tile: the 32 diagonal sums it needs first (into LDS), then its elements.  The
// distance partials: every tile leaves its sum in scratch[], the last tile to arrive (counter) adds them -- tile q into part
// q mod 32, in ascending q -- so the result does not depend on the order the tiles ran in.
__global__ __launch_bounds__(256) void splitk_reduce_sym_kernel(GemmArgs g, int nsplit, int mode, double *__restrict__ dinv,
                                                                double *__restrict__ dist, double *__restrict__ scratch,
                                                                int *__restrict__ counter) {
  __shared__ double dg[32];
  __shared__ double red[256];
  __shared__ int last;
  const int b = g.M, tid = threadIdx.x;
  const int nt1 = (b + 15) / 16;
  const int i0 = (blockIdx.x / nt1) * 16, j0 = (blockIdx.x % nt1) * 16;
  // the planes of this thread's element and -- threads 0..31 -- of one diagonal element, four loads of each in flight
  const int i = i0 + (tid >> 4), j = j0 + (tid & 15);
  const bool live = i < b && j < b;
  const int q = (tid < 16) ? i0 + tid : j0 + tid - 16;
  const bool hasd = (mode & 1) && tid < 32 && q < b;
  const size_t plane = (size_t)b * b;
  const double *pe = g.part + (live ? (size_t)i * b + j : 0);
  const double *pd = g.part + (hasd ? (size_t)q * b + q : 0);
  double v = 0.0, dsum = 0.0;
  int z = 0;
  for (; z + 4 <= nsplit; z += 4) {
    double e[4], dq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      e[u] = live ? pe[(size_t)(z + u) * plane] : 0.0;
      dq[u] = hasd ? pd[(size_t)(z + u) * plane] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { v += e[u]; dsum += dq[u]; }      // fixed order
  }
  for (; z < nsplit; ++z) {
    if (live) v += pe[(size_t)z * plane];
    if (hasd) dsum += pd[(size_t)z * plane];
  }
  if ((mode & 1) && tid < 32) dg[tid] = dsum > 0.0 ? 1.0 / __builtin_sqrt(dsum) : 0.0;
  __syncthreads();
  double d2 = 0.0;
  if (live) {
    // (row, column) of the caller's matrix: whichever of (i, j) has the unit stride is the row of a column-major result
    const bool j_is_row = g.c_js == 1;
    const int row = j_is_row ? j : i, col = j_is_row ? i : j;
    if (mode & 1) {
      const double drow = j_is_row ? dg[16 + (tid & 15)] : dg[tid >> 4], dcol = j_is_row ? dg[tid >> 4] : dg[16 + (tid & 15)];
      v = v * drow * dcol;
      if (i == j) dinv[i] = dg[tid >> 4];
    }
    if ((mode & 2) && row >= col) v = 0.0;
    if ((mode & 8) && row <= col) v = 0.0;
    g.C[(size_t)i * g.c_is + (size_t)j * g.c_js] = v;
    const double df = v - (i == j ? 1.0 : 0.0);
    d2 = df * df;
  }
  if (!(mode & 4)) return;
  red[tid] = d2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    scratch[blockIdx.x] = red[0];
    __threadfence();
    last = (atomicAdd(counter, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (tid < GEMM_DIST_PARTS) {
    double a = 0.0;
    for (int q = tid; q < (int)gridDim.x; q += GEMM_DIST_PARTS)   // device-scope loads: the sums were written through other XCDs' L2
      a += __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)&scratch[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    dist[tid] = a;
    __threadfence_system();   // (dist may be host memory)
  }
  if (tid == 0) *counter = 0;
}

int gemm_launch(hipStream_t st, int M, int N, int Kd, double alpha, const double *A, long a_is, long a_ks,
                const double *B, long b_ks, long b_js, double beta, const double *E, long e_is, long e_js,
                double *C, long c_is, long c_js, double *work, size_t work_elems, double gamma,
                const double *E2, int *tickets, GemmFusedReduce *fused, const GemmPair *pair) {
  if (fused) fused->done = false;
  if (pair && (E || E2 || fused)) { set_error("gemm: a paired product takes no E / E2 / fused reduction"); return FLGP_ERR_INVALID; }
  if (M <= 0 || N <= 0) return FLGP_OK;
  GemmArgs g;
  // orient so that the contiguous output dimension is the kernel's column dimension
  if (c_is == 1 && c_js != 1) {
    g.M = N; g.N = M; g.Kd = Kd;
    g.A = B; g.a_is = b_js; g.a_ks = b_ks;
    g.B = A; g.b_ks = a_ks; g.b_js = a_is;
    g.E = E; g.e_is = e_js; g.e_js = e_is;
    g.C = C; g.c_is = c_js; g.c_js = c_is;
    if (pair) { g.A2 = pair->B2; g.B2 = pair->A2; }
  } else {
    g.M = M; g.N = N; g.Kd = Kd;
    g.A = A; g.a_is = a_is; g.a_ks = a_ks;
    g.B = B; g.b_ks = b_ks; g.b_js = b_js;
    g.E = E; g.e_is = e_is; g.e_js = e_js;
    g.C = C; g.c_is = c_is; g.c_js = c_js;
    if (pair) { g.A2 = pair->A2; g.B2 = pair->B2; }
  }
  if (!pair) { g.A2 = g.B2 = nullptr; g.C2 = nullptr; } else g.C2 = pair->C2;
  g.tickets = (work && !fused && !pair && tuning("gemm_fused_reduce", 0)) ? tickets : nullptr;
  g.alpha = alpha; g.beta = beta; g.gamma = gamma;
  g.E2 = (gamma == 0.0) ? nullptr : E2;
  if (beta == 0.0) g.E = nullptr;
  // tile size: 64 where tiles of 128 would leave most of the 256 CUs without one (the solver's s x b and b x b products)
  int gbt = GB;
  if (ceil_div(g.M, GB) * ceil_div(g.N, GB) < tuning("gemm_tile64_below", 200) && !g.tickets) gbt = 64;
  const int ntiles = ceil_div(g.M, gbt) * ceil_div(g.N, gbt);
  if (ntiles > GEMM_MAX_TICKETS) g.tickets = nullptr;
  // one partial plane: M x N doubles for the reduction kernel, whole 128 x 128 tiles for the in-kernel reduction
  const size_t per = g.tickets ? (size_t)ntiles * PLANE : (size_t)g.M * g.N;
  int nsplit = 1;
  if (work && !pair && ntiles < 256 && Kd >= 8 * GK) {
    nsplit = (gbt == 64 ? tuning("gemm_tile64_blocks", 256) : 512) / ntiles;
    // at least gemm_min_stages (default 5) stages per block: with fewer, the partial planes (and the reduction that
    // reads them back) cost more than the extra blocks gain -- the 256 x 256 x 5000 Gram products of the
    // eigensolver spent 33 us in the reduction of 128 planes next to 26 us in the GEMM
    const int maxk = Kd / (tuning("gemm_min_stages", 5) * GK);
    if (nsplit > maxk) nsplit = maxk;
    if ((size_t)nsplit * per > work_elems) nsplit = (int)(work_elems / per);
    if (nsplit < 1) nsplit = 1;
  }
  int klen = ceil_div(Kd > 0 ? Kd : 1, nsplit);
  klen = (klen + GK - 1) / GK * GK;
  nsplit = ceil_div(Kd > 0 ? Kd : 1, klen);
  g.ksplit_len = klen;
  g.part = work;
  // overlapping tiles write some elements twice: harmless unless the epilogue reads what it overwrites
  // (with the reduction kernel every element is written exactly once, by that kernel, whatever the tiles overlap)
  const bool aliased = (const double *)g.C == g.E || (const double *)g.C == g.E2;
  g.shift_edges = ((nsplit > 1 && !g.tickets) || !aliased) ? 1 : 0;
  {
    const double fl = 2.0 * (double)M * (double)N * (double)Kd;
    ProfScope ps("gemm_f64_kernel", st, fl);
    // second record per shape class (large / medium / small) for the bench breakdown
    ProfScope ps2(fl > 5e9 ? "gemm_large" : (fl > 2e8 ? "gemm_medium" : "gemm_small"), st, fl);
    const int ny = pair ? 2 : 1;
    // unused dynamic LDS caps the workgroups the dispatcher may stack on one CU (experiment knob, KB)
    const int pad_kb = (gbt == 64) ? tuning("gemm_lds_pad_kb", 0) : 0;
    if (pad_kb > 0) {
      static int attr_set = 0;
      if (!attr_set) { FLGP_HIP(hipFuncSetAttribute((const void *)gemm_f64_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024)); attr_set = 1; }
    }
    if (gbt == 64) hipLaunchKernelGGL(gemm_f64_kernel<64>, dim3(ntiles, ny, nsplit), dim3(256), (size_t)pad_kb * 1024, st, g);
    else hipLaunchKernelGGL(gemm_f64_kernel<128>, dim3(ntiles, ny, nsplit), dim3(256), 0, st, g);
  }
  FLGP_TRY(check_launch("gemm_f64_kernel"));
  if (!g.tickets && nsplit > 1) {
    if (fused && g.M == g.N && alpha == 1.0 && !g.E && !g.E2 && (g.c_is == 1 || g.c_js == 1)) {
      const int nt1 = ceil_div(g.M, 16);
      hipLaunchKernelGGL(splitk_reduce_sym_kernel, dim3(nt1 * nt1), dim3(256), 0, st, g, nsplit, fused->mode, fused->dinv,
                         fused->dist, fused->scratch, fused->counter);
      FLGP_TRY(check_launch("splitk_reduce_sym_kernel"));
      fused->done = true;
      return FLGP_OK;
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div((long)g.M * g.N, 256)), dim3(256), 0, st, g, nsplit);
    FLGP_TRY(check_launch("splitk_reduce_kernel"));
  }
  return FLGP_OK;
}

// Vw(b,k) = exp(-t (1 - values_k)) * V1(row(b), k), stored b-contiguous: Vw[b + k*n1]
__global__ void hk_scale_kernel(const double *__restrict__ values, int K, double t, const double *__restrict__ V1,
                                int ld1, const int *__restrict__ idx1, int row0, int n1, double *__restrict__ Vw) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)n1 * K) return;
  const int b = (int)(e % n1), k = (int)(e / n1);
  const int row = idx1 ? idx1[b] : row0 + b;
  const double w = exp(-t * (1.0 - values[k]));  // src/Spectrum.cpp:86,90
  Vw[e] = V1[(size_t)k * ld1 + row] * w;
}

__global__ void gather_rows_kernel(const double *__restrict__ V, int ld, const int *__restrict__ idx, int n0, int K,
                                   double *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)n0 * K) return;
  const int a = (int)(e % n0), k = (int)(e / n0);
  out[e] = V[(size_t)k * ld + idx[a]];
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_gemm(void *stream, int M, int N, int Kd, double alpha, const double *A, long a_is,
                             long a_ks, const double *B, long b_ks, long b_js, double beta, const double *E,
                             long e_is, long e_js, double *C, long c_is, long c_js, double *d_work,
                             size_t work_elems) {
  FLGP_REQUIRE(M >= 0 && N >= 0 && Kd >= 0 && A && B && C, "gemm: bad arguments");
  return gemm_launch((hipStream_t)stream, M, N, Kd, alpha, A, a_is, a_ks, B, b_ks, b_js, beta, E, e_is, e_js, C,
                     c_is, c_js, d_work, work_elems, 0.0, nullptr);
}

// The reduction half of a split product on its own (rot.hip's Gram kernel writes the planes): C(ic, jc) = sum over planes of
// part[z][jc][ic] for a column-major b x b result with leading dimension b, with the fused extras of GemmFusedReduce.
namespace flgp {
int gemm_reduce_square(hipStream_t st, int b, const double *part, int nsplit, double *C, GemmFusedReduce *fused) {
  GemmArgs g;
  g.M = b; g.N = b; g.Kd = 0;
  g.A = g.B = nullptr; g.a_is = g.a_ks = g.b_ks = g.b_js = 0;
  g.alpha = 1.0; g.beta = 0.0; g.gamma = 0.0;
  g.E = g.E2 = nullptr; g.e_is = g.e_js = 0;
  g.C = C; g.c_is = b; g.c_js = 1;            // kernel (i, j) = caller (column, row), as gemm_launch orients a column-major result
  g.shift_edges = 0; g.ksplit_len = 0; g.part = const_cast<double *>(part); g.tickets = nullptr;
  g.A2 = g.B2 = nullptr; g.C2 = nullptr;
  if (fused) {
    const int nt1 = ceil_div(b, 16);
    hipLaunchKernelGGL(splitk_reduce_sym_kernel, dim3(nt1 * nt1), dim3(256), 0, st, g, nsplit, fused->mode, fused->dinv, fused->dist,
                       fused->scratch, fused->counter);
    fused->done = true;
    return check_launch("splitk_reduce_sym_kernel");
  }
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div((long)b * b, 256)), dim3(256), 0, st, g, nsplit);
  return check_launch("splitk_reduce_kernel");
}
}  // namespace flgp

extern "C" int flgp_dev_gemm_pair(void *stream, int M, int N, int Kd, double alpha, const double *A, const double *A2, long a_is,
                                  long a_ks, const double *B, const double *B2, long b_ks, long b_js, double *C, double *C2,
                                  long c_is, long c_js) {
  FLGP_REQUIRE(M >= 0 && N >= 0 && Kd >= 0 && A && B && C && A2 && B2 && C2, "gemm_pair: bad arguments");
  const GemmPair pr{A2, B2, C2};
  return gemm_launch((hipStream_t)stream, M, N, Kd, alpha, A, a_is, a_ks, B, b_ks, b_js, 0.0, nullptr, 0, 0, C, c_is, c_js,
                     nullptr, 0, 0.0, nullptr, nullptr, nullptr, &pr);
}

extern "C" int flgp_dev_gather_rows(void *stream, const double *dV, int ld, const int *d_idx, int n0, int K,
                                    double *d_out) {
  FLGP_REQUIRE(dV && d_idx && d_out && n0 >= 0 && K >= 1, "gather_rows: bad arguments");
  if (n0 == 0) return FLGP_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long)n0 * K, 256)), dim3(256), 0, (hipStream_t)stream, dV, ld, d_idx,
                     n0, K, d_out);
  return check_launch("gather_rows_kernel");
}

extern "C" size_t flgp_dev_hk_workspace(int n0, int n1, int K, int gather0) {
  // (the padded operand of the panel kernel, hk.hip, is the larger of the two layouts of Vw)
  return sizeof(double) * (hk_panel_vw_elems(n1, K) + (gather0 ? (size_t)n0 * K : 0)) + 256;
}

extern "C" int flgp_dev_hk(void *stream, const double *d_values, int K, double t, const double *dV0, int ld0,
                           const int *d_idx0, int row0_0, int n0, const double *dV1, int ld1, const int *d_idx1,
                           int row0_1, int n1, double *dH, int ldh, double *d_work) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(K >= 1 && n0 >= 0 && n1 >= 0 && ldh >= n0, "HK: bad shape");
  if (n0 == 0 || n1 == 0) return FLGP_OK;
  double *Vw = d_work;
  const double *V0 = dV0 + row0_0;
  long v0_ld = ld0;
  if (d_idx0) {  // general row gather (mat_indexing, src/Utils.h:130-137); callers normally pass ranges
    double *G0 = d_work + hk_panel_vw_elems(n1, K);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long)n0 * K, 256)), dim3(256), 0, st, dV0, ld0, d_idx0,
                       n0, K, G0);
    FLGP_TRY(check_launch("gather_rows_kernel"));
    V0 = G0;
    v0_ld = n0;
  }
  // the path's own shape (many rows of V against the training block): panels of V resident in LDS, hk.hip
  if (hk_panel2_applicable(n0, n1, K, ldh))
    return hk_panel2_launch(st, d_values, K, t, V0, v0_ld, n0, dV1, ld1, d_idx1, row0_1, n1, dH, ldh, Vw);
  if (hk_panel_applicable(n0, n1, K, ldh))
    return hk_panel_launch(st, d_values, K, t, V0, v0_ld, n0, dV1, ld1, d_idx1, row0_1, n1, dH, ldh, Vw);
  hipLaunchKernelGGL(hk_scale_kernel, dim3(ceil_div((long)n1 * K, 256)), dim3(256), 0, st, d_values, K, t, dV1, ld1,
                     d_idx1, row0_1, n1, Vw);
  FLGP_TRY(check_launch("hk_scale_kernel"));
  // H(a,b) = sum_k V0(a,k) Vw(b,k)
  return gemm_launch(st, n0, n1, K, 1.0, V0, 1, v0_ld, Vw, n1, 1, 0.0, nullptr, 0, 0, dH, 1, ldh, nullptr, 0, 0.0,
                     nullptr);
}
