// k7, second version of the LDS-panel contraction (see hk.hip for the idea): the issue stream cut down to what the
// matrix pipe tolerates.
//
// Measured on the box (scripts/ubench_issue.hip): next to a stream of v_mfma_f64_16x16x4_f64 (64 cycles each), EVERY other
// instruction a SIMD issues -- vector, scalar, LDS or memory, from either of its two waves -- costs the matrix pipe
// about four cycles.  hk.hip's loop carries ~20 of them per k step (run-time LDS indices, 64-bit address arithmetic for
// its register ring, the flattened stage / tile bookkeeping and its branches): (4 x 64 + 80) / (4 x 64) = 76 % busy,
// which is what the counters say.  Here:
//   * a wave owns TWO m-tiles at a time (8 MFMAs per k step share the same four B fragments: half the LDS reads per MFMA);
//   * the k loop of a tile is unrolled completely (the stage count is a template parameter), so every LDS fragment read
//     is `ds_read2st64_b64` with immediate offsets off one base register -- no index arithmetic, no branch;
//   * the small operand is stored SWIZZLED by hk2_scale_kernel, [tile pair][k-step pair][tile][lane][2]: the A
//     fragments of two k steps of one tile are one 16-byte load per lane, 1 KB contiguous per wave, scalar base +
//     one 32-bit lane offset; a ring of two slots (two k steps each, i.e. 32 MFMAs of look-ahead per slot) divides every
//     stage count, so the ring runs on across tiles without a remainder;
//   * a tile ends with 32 stores through scalar bases; the first and the last tile of a panel have their own copies of
//     the body (the last one sends for the next panel), so that at every loop head the queue of outstanding memory
//     operations is the same on all incoming edges and the compiler's `s_waitcnt vmcnt` counts are exact -- a wave
//     never waits for its own stores.
// ~5 other instructions per 8 MFMAs.  Arithmetic unchanged: per element the k-ascending MFMA chain from zero, bit for
// bit the tiled GEMM's H (tests/test_gpu_parity.py::test_hk_panel_kernel_bit_identical_to_gemm).
#include "common.h"
#include <type_traits>

namespace flgp {

typedef double kd4 __attribute__((ext_vector_type(4)));
typedef double kd2 __attribute__((ext_vector_type(2)));
typedef unsigned int ku4 __attribute__((ext_vector_type(4)));
typedef unsigned int ku2 __attribute__((ext_vector_type(2)));
// Buffer addressing (scalar resource + scalar offset + one 32-bit lane offset + immediate) for the operand stream and
// the H stores: with flat 64-bit addresses the compiler keeps one VGPR pair per unrolled load (it hoists
// base + lane + constant out of the tile loop: 100 + registers, spills), and every address costs VALU issue slots.
constexpr unsigned BUF_WORD3 = 0x00020000u;      // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)

constexpr int HP2 = 64;             // rows of V0 per panel
constexpr int HK2_WAVES = 8;

struct Hk2Args {
  const double *V0; long ld0; int n0;     // V0(a, k) = V0[a + k ld0], a < n0
  const double *Vsw; int n1, npairs;      // swizzled small operand: npairs pairs of m-tiles (rows >= n1 and k >= K are zero)
  unsigned vsw_bytes;
  int K;
  double *H; long ldh;                    // H(a, b) = H[a + b ldh]
  int nblocks;                            // panels: ceil(n0 / HP2)
};

template <int NST, int PG, int STORE_AUX>
__global__ __launch_bounds__(512, 1) void hk_panel2_kernel(Hk2Args g) {
  constexpr int NKS = NST * 4;            // k steps per tile
  constexpr int NKP = NST * 2;            // k-step pairs = ring stages per tile
  constexpr size_t PAIR_BYTES = (size_t)NKP * 2048;     // one tile pair of the swizzled operand
  extern __shared__ double panel[];       // [NKS][HP2][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int ntw = (g.npairs - wave + HK2_WAVES - 1) / HK2_WAVES;   // tile pairs of this wave: wave, wave + 8, ...

  // ---- panel transport (as in hk.hip): wave pg0 takes the k groups pg0, pg0 + 8, ...; lane = row
  const int pj = lane, pg0 = wave;
  double pr[PG][4];
  auto panel_fetch = [&](int blk) {
    long a = (long)blk * HP2 + pj;
    if (a > (long)g.n0 - 1) a = (long)g.n0 - 1;
    const unsigned aoff = (unsigned)(a - (long)blk * HP2) * 8u;
    const double *src = g.V0 + (size_t)blk * HP2;
#pragma unroll
    for (int q = 0; q < PG; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int k = 4 * (pg0 + 8 * q) + c;
        if (k > g.K - 1) k = g.K - 1;
        const double *col = src + (size_t)k * g.ld0;
        pr[q][c] = *(const double *)((const char *)col + aoff);
      }
    }
  };
  auto panel_put = [&]() {
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int kg = pg0 + 8 * q;
      if (kg < NKS) {
        kd2 lo, hi;
        lo[0] = (4 * kg + 0 < g.K) ? pr[q][0] : 0.0;
        lo[1] = (4 * kg + 1 < g.K) ? pr[q][1] : 0.0;
        hi[0] = (4 * kg + 2 < g.K) ? pr[q][2] : 0.0;
        hi[1] = (4 * kg + 3 < g.K) ? pr[q][3] : 0.0;
        kd2 *dst = (kd2 *)(panel + ((size_t)kg * HP2 + pj) * 4);
        dst[0] = lo;
        dst[1] = hi;
      }
    }
  };

  int blk = blockIdx.x;
  if (blk < g.nblocks) panel_fetch(blk);
  const unsigned a_lane = (unsigned)lane * 16u;                          // lane part of an A address (bytes)
  const unsigned h_lane = ((unsigned)fr + (unsigned)fk * (unsigned)g.ldh) * 8u;
  const double *pan_lane = panel + (size_t)fr * 4 + fk;
  const __amdgpu_buffer_rsrc_t vsw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Vsw, 0, (int)g.vsw_bytes, BUF_WORD3);
  const unsigned vsw_wave = (unsigned)wave * (unsigned)PAIR_BYTES;       // byte offset of this wave's first tile pair
  unsigned row_off[2][4];                                                // byte offsets of the 8 row groups of a tile pair in H
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) row_off[m][reg] = (unsigned)(m * 16 + 4 * reg) * (unsigned)g.ldh * 8u;

  for (; blk < g.nblocks; blk += gridDim.x) {
    panel_put();
    __syncthreads();
    const int nxt = blk + (int)gridDim.x;
    const bool has_next = nxt < g.nblocks;
    if (ntw == 0 && has_next) panel_fetch(nxt);
    const bool cols_inside = (long)blk * HP2 + HP2 <= (long)g.n0;

    kd2 ar[2][2];                 // [ring slot][m-tile]: the A fragments of two k steps
    kd4 acc[2][4];
    double bf[4];
    auto a_fetch = [&](kd2 (&dst)[2], unsigned so) {   // the stage at byte offset so (uniform) of the swizzled operand
      dst[0] = __builtin_bit_cast(kd2, __builtin_amdgcn_raw_buffer_load_b128(vsw_rsrc, a_lane, so, 0));
      dst[1] = __builtin_bit_cast(kd2, __builtin_amdgcn_raw_buffer_load_b128(vsw_rsrc, a_lane + 1024u, so, 0));
    };
    // one tile pair: NKP stages of two k steps; the loads of stage kp + 2 follow stage kp (the first two stages of the
    // NEXT pair behind the last two of this one)
    auto tile = [&](int t, auto first_c, auto last_c) {
      constexpr bool LAST = decltype(last_c)::value;
      const unsigned cur = vsw_wave + (unsigned)t * (unsigned)(HK2_WAVES * PAIR_BYTES);
      const unsigned nx = LAST ? cur : cur + (unsigned)(HK2_WAVES * PAIR_BYTES);   // (the last pair re-reads itself: unused)
      if constexpr (LAST) { if (has_next) panel_fetch(nxt); }
      unsigned so_run = cur + 2048u;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[m][ni] = kd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          constexpr int dummy = 0; (void)dummy;
          const int ks = 2 * kp + j;
          const int kgn = (ks + 1 < NKS) ? ks + 1 : 0;
          double bn[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bn[ni] = pan_lane[(size_t)kgn * (HP2 * 4) + ni * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[m][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[kp & 1][m][j], bf[ni], acc[m][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
        }
        // offset of stage kp + 2: one scalar add where it is needed (as C++ arithmetic the scheduler computes all 26 of a
        // tile at its top, where they overflow the SGPR file into VGPR lanes)
        if (kp + 2 == NKP) so_run = nx;
        else asm volatile("s_add_u32 %0, %0, 0x800" : "+s"(so_run) : : "scc");
        a_fetch(ar[kp & 1], so_run);
      }
      // ---- both m-tiles are complete: D(row = fk + 4 reg, col = fr) of each 16 x 16 tile
      const int bt = (2 * (wave + HK2_WAVES * t)) * 16;               // first row of the pair (uniform)
      // (a resource per tile pair: its 32 rows of H span 35 ldh doubles, which 32-bit offsets reach; all of H they do not)
      const __amdgpu_buffer_rsrc_t h_rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)(g.H + (size_t)blk * HP2 + (size_t)bt * g.ldh), 0, 0x7fffffff, BUF_WORD3);
      if (cols_inside && bt + 32 <= g.n1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const double v = acc[m][ni][reg];      // (a copy: __builtin_bit_cast of the vector element itself took element 0)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ku2, v), h_rsrc, h_lane + (unsigned)(ni * 128), row_off[m][reg], STORE_AUX);
            }
          }
      } else {
        const long a0 = (long)blk * HP2 + fr;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int b = bt + m * 16 + 4 * reg + fk;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              if (a0 + ni * 16 < (long)g.n0 && b < g.n1) {
                const double v = acc[m][ni][reg];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ku2, v), h_rsrc, h_lane + (unsigned)(ni * 128), row_off[m][reg], STORE_AUX);
              }
          }
      }
    };
    typedef std::integral_constant<bool, true> T_;
    typedef std::integral_constant<bool, false> F_;
    if (ntw > 0) {
      a_fetch(ar[0], vsw_wave);
      a_fetch(ar[1], vsw_wave + 2048u);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = pan_lane[ni * 64];
      if (ntw == 1) {
        tile(0, T_{}, T_{});
      } else {
        tile(0, T_{}, F_{});
        for (int t = 1; t + 1 < ntw; ++t) tile(t, F_{}, F_{});
        tile(ntw - 1, F_{}, T_{});
      }
    }
    __syncthreads();     // every wave has finished with the panel
  }
}

// Swizzled small operand: Vsw[((pair * NKP + kp) * 2 + m) * 128 + lane * 2 + j] = Vw(b, k) with
//   b = (2 pair + m) * 16 + (lane & 15),  k = (2 kp + j) * 4 + (lane >> 4),
//   Vw(b, k) = exp(-t (1 - values_k)) * V1(row(b), k)  for b < n1, k < K, else 0      (src/Spectrum.cpp:86,90)
__global__ void hk2_scale_kernel(const double *__restrict__ values, int K, int nkp, double t,
                                 const double *__restrict__ V1, int ld1, const int *__restrict__ idx1, int row0, int n1,
                                 int npairs, double *__restrict__ Vsw) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)npairs * nkp * 256;
  if (e >= tot) return;
  const int j = (int)(e & 1), lane = (int)((e >> 1) & 63), m = (int)((e >> 7) & 1);
  const long r = e >> 8;
  const int kp = (int)(r % nkp), pair = (int)(r / nkp);
  const int b = (2 * pair + m) * 16 + (lane & 15), k = (2 * kp + j) * 4 + (lane >> 4);
  double v = 0.0;
  if (b < n1 && k < K) {
    const int row = idx1 ? idx1[b] : row0 + b;
    const double w = exp(-t * (1.0 - values[k]));
    v = V1[(size_t)k * ld1 + row] * w;
  }
  Vsw[e] = v;
}

bool hk_panel2_applicable(int n0, int n1, int K, long ldh) {
  const int nst = (K + 15) / 16;
  return tuning("hk_panel", 1) && tuning("hk_panel2", 1) && (nst == 13 || nst == 7) && n1 >= tuning("hk_panel2_min_n1", 480) &&    // (fewer than 15 tile pairs leave waves without work)
         n0 >= tuning("hk_panel_min_n0", 2048) && ldh <= 7000000L && n1 <= 100000;   // (store offsets of a tile pair, (35 ldh + 63) * 8, and the operand's size stay below 2^31)
}

int hk_panel2_launch(hipStream_t st, const double *d_values, int K, double t, const double *V0, long ld0, int n0,
                     const double *dV1, int ld1, const int *d_idx1, int row0_1, int n1, double *dH, long ldh,
                     double *d_vw) {
  const int nst = (K + 15) / 16, nkp = nst * 2, npairs = (n1 + 31) / 32;
  hipLaunchKernelGGL(hk2_scale_kernel, dim3(ceil_div((long)npairs * nkp * 256, 256)), dim3(256), 0, st, d_values, K, nkp, t,
                     dV1, ld1, d_idx1, row0_1, n1, npairs, d_vw);
  FLGP_TRY(check_launch("hk2_scale_kernel"));
  Hk2Args g;
  g.V0 = V0; g.ld0 = ld0; g.n0 = n0;
  g.Vsw = d_vw; g.n1 = n1; g.npairs = npairs;
  g.vsw_bytes = (unsigned)((size_t)npairs * nkp * 256 * sizeof(double));
  g.K = K;
  g.H = dH; g.ldh = ldh;
  g.nblocks = ceil_div(n0, HP2);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n_cu = v;
  }
  int grid = n_cu;
  if (grid > g.nblocks) grid = g.nblocks;
  const size_t lds = sizeof(double) * (size_t)nst * 16 * HP2;
  const double fl = 2.0 * (double)n0 * (double)n1 * (double)K;
  ProfScope ps("hk_panel_kernel", st, fl);
  auto go = [&](auto kfn) -> int {
    FLGP_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, g);
    return FLGP_OK;
  };
  // H is written once and not read again on the device: non-temporal stores (aux bit 1) keep it from pushing the
  // L2-resident small operand and the next panels out of the caches (knob hk_store_nt, default on: see DESIGN.md)
  const bool nt = tuning("hk_store_nt", 1) != 0;
  if (nst == 13) { if (nt) FLGP_TRY(go(hk_panel2_kernel<13, 7, 2>)); else FLGP_TRY(go(hk_panel2_kernel<13, 7, 0>)); }
  else { if (nt) FLGP_TRY(go(hk_panel2_kernel<7, 4, 2>)); else FLGP_TRY(go(hk_panel2_kernel<7, 4, 0>)); }
  return check_launch("hk_panel2_kernel");
}

}  // namespace flgp
