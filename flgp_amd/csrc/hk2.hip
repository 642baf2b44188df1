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

// Round 4.  (1) The stage count follows K exactly: NKS k steps (K = 200: 50, not 52 -- 4 % of the MFMAs were multiplications
// by the zero padding); a ring stage is two k steps, so NKP = NKS / 2 may be odd (25) and the two-slot ring then changes
// parity from one tile to the next: the tile body takes the parity as a template parameter.  (2) The next panel no longer
// waits in 56 VGPRs through the whole last tile (484 bytes of scratch per lane, re-written every panel: 1.9 GB of HBM
// writes per launch that were not H): LDS holds TWO copies of the panel's first NA k steps (A0 / A1, used alternately) and
// one of the remaining NB = NKS - NA (B) -- (2 NA + NB) x 2 KB <= 156 KB.  During a panel's last tile the next panel's A part
// goes through four short-lived register chunks (4 doubles each) straight into the idle A copy; only its B part (NB / 2
// doubles per lane) is kept in registers until the barrier that frees B.  The fetch is unconditional (the last panel fetches
// itself again) so that every path into a tile has the same queue of outstanding loads and the compiler's vmcnt counts
// stay exact.
template <int NKS, int NA, int STORE_AUX>
__global__ __launch_bounds__(512, 1) void hk_panel2_kernel(Hk2Args g) {
  static_assert(NKS % 2 == 0 && NA <= NKS && NA >= 8, "k steps come in ring stages of two");
  constexpr int NKP = NKS / 2;            // ring stages per tile
  constexpr int NB = NKS - NA;
  constexpr int QA = (NA + 7) / 8, QB = (NB + 7) / 8;   // k steps of a part per wave (wave w takes w, w + 8, ...)
  constexpr bool ODD = (NKP & 1) != 0;
  constexpr size_t PAIR_BYTES = (size_t)NKP * 2048;     // one tile pair of the swizzled operand
  constexpr int SLOT = HP2 * 4;                          // doubles of one k step in LDS: [row][k % 4]
  extern __shared__ double panel[];       // A0 [NA][HP2][4] | A1 [NA][HP2][4] | B [NB][HP2][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int ntw = (g.npairs - wave + HK2_WAVES - 1) / HK2_WAVES;   // tile pairs of this wave: wave, wave + 8, ...

  // ---- panel transport: lane = row of the panel; a k step is four values of one row = 32 contiguous bytes in LDS
  const int pj = lane;
  auto fetch_ks = [&](int blk, int ks, double (&dst)[4]) {     // k step ks of panel blk, this lane's row (clamped; put_ks zeroes k >= K)
    long a = (long)blk * HP2 + pj;
    if (a > (long)g.n0 - 1) a = (long)g.n0 - 1;
    const unsigned aoff = (unsigned)(a - (long)blk * HP2) * 8u;
    const double *src = g.V0 + (size_t)blk * HP2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int k = 4 * ks + c;
      if (k > g.K - 1) k = g.K - 1;
      const double *col = src + (size_t)k * g.ld0;
      dst[c] = *(const double *)((const char *)col + aoff);
    }
  };
  auto put_ks = [&](double *slot0, int ks, const double (&v)[4]) {   // slot0: first k step of the LDS region, ks relative to it is added by the caller
    kd2 lo, hi;
    lo[0] = (4 * ks + 0 < g.K) ? v[0] : 0.0;
    lo[1] = (4 * ks + 1 < g.K) ? v[1] : 0.0;
    hi[0] = (4 * ks + 2 < g.K) ? v[2] : 0.0;
    hi[1] = (4 * ks + 3 < g.K) ? v[3] : 0.0;
    kd2 *dst = (kd2 *)(slot0 + (size_t)pj * 4);
    dst[0] = lo;
    dst[1] = hi;
  };
  double *const regB = panel + (size_t)2 * NA * SLOT;
  double prB[QB > 0 ? QB : 1][4];
  auto fetch_B = [&](int blk) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      int ks = NA + wave + 8 * q;
      if (ks > NKS - 1) ks = NKS - 1;          // (a wave without a q-th k step re-reads the last one; put_B skips it)
      fetch_ks(blk, ks, prB[q]);
    }
  };
  auto put_B = [&]() {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int ks = NA + wave + 8 * q;
      if (ks < NKS) put_ks(regB + (size_t)(ks - NA) * SLOT, ks, prB[q]);
    }
  };

  int blk = blockIdx.x;
  const int last_blk = g.nblocks - 1;
  int par = 0;                                  // which A copy holds the current panel
  if (blk < g.nblocks) {                        // prologue: the first panel, A part through registers as well
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int ks = wave + 8 * q;
      double t4[4];
      fetch_ks(blk, ks < NA ? ks : NA - 1, t4);
      if (ks < NA) put_ks(panel + (size_t)ks * SLOT, ks, t4);
    }
    fetch_B(blk);
  }
  const unsigned a_lane = (unsigned)lane * 16u;                          // lane part of an A address (bytes)
  const unsigned h_lane = ((unsigned)fr + (unsigned)fk * (unsigned)g.ldh) * 8u;
  const double *const panB = regB + (size_t)fr * 4 + fk;
  const __amdgpu_buffer_rsrc_t vsw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Vsw, 0, (int)g.vsw_bytes, BUF_WORD3);
  const unsigned vsw_wave = (unsigned)wave * (unsigned)PAIR_BYTES;       // byte offset of this wave's first tile pair
  unsigned row_off[2][4];                                                // byte offsets of the 8 row groups of a tile pair in H
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) row_off[m][reg] = (unsigned)(m * 16 + 4 * reg) * (unsigned)g.ldh * 8u;

  for (; blk < g.nblocks; blk += gridDim.x) {
    put_B();
    __syncthreads();
    int nxt = blk + (int)gridDim.x;
    if (nxt > last_blk) nxt = last_blk;         // (unconditional fetch: the last panel of a workgroup fetches a valid panel again)
    const double *const panA = panel + (size_t)par * NA * SLOT + (size_t)fr * 4 + fk;
    double *const nextA = panel + (size_t)(par ^ 1) * NA * SLOT;
    const bool cols_inside = (long)blk * HP2 + HP2 <= (long)g.n0;

    kd2 ar[2][2];                 // [ring slot][m-tile]: the A fragments of two k steps
    kd4 acc[2][4];
    double bf[4];
    double chunk[4];              // the next panel's A part on its way into LDS, one k step at a time
    auto a_fetch = [&](kd2 (&dst)[2], unsigned so) {   // the stage at byte offset so (uniform) of the swizzled operand
      dst[0] = __builtin_bit_cast(kd2, __builtin_amdgcn_raw_buffer_load_b128(vsw_rsrc, a_lane, so, 0));
      dst[1] = __builtin_bit_cast(kd2, __builtin_amdgcn_raw_buffer_load_b128(vsw_rsrc, a_lane + 1024u, so, 0));
    };
    // one tile pair: NKP stages of two k steps; the loads of stage kp + 2 follow stage kp (the first two stages of the
    // NEXT pair behind the last two of this one).  PAR: the ring slot of stage 0.
    auto tile = [&](int t, auto last_c, auto par_c) {
      constexpr bool LAST = decltype(last_c)::value;
      constexpr int PAR = decltype(par_c)::value;
      const unsigned cur = vsw_wave + (unsigned)t * (unsigned)(HK2_WAVES * PAIR_BYTES);
      const unsigned nx = LAST ? cur : cur + (unsigned)(HK2_WAVES * PAIR_BYTES);   // (the last pair re-reads itself: unused)
      unsigned so_run = cur + 2048u;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[m][ni] = kd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        if constexpr (LAST) {
          // the next panel's A part: chunk q is loaded at stage SP q and stored at stage SP (q + 1) -- five stages are ~2.5 us
          constexpr int SP = (NKP - 1) / (QA + 1) > 0 ? (NKP - 1) / (QA + 1) : 1;
          if (kp % SP == 0 && kp / SP >= 1 && kp / SP <= QA) {
            const int q = kp / SP - 1, ks = wave + 8 * q;
            if (ks < NA) put_ks(nextA + (size_t)ks * SLOT, ks, chunk);
          }
          if (kp % SP == 0 && kp / SP < QA) {
            const int q = kp / SP, ks = wave + 8 * q;
            fetch_ks(nxt, ks < NA ? ks : NA - 1, chunk);
          }
          if (kp == SP * QA) fetch_B(nxt);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ks = 2 * kp + j;
          const int kgn = (ks + 1 < NKS) ? ks + 1 : 0;
          double bn[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            bn[ni] = (kgn < NA) ? panA[(size_t)kgn * SLOT + ni * 64] : panB[(size_t)(kgn - NA) * SLOT + ni * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[m][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[(kp + PAR) & 1][m][j], bf[ni], acc[m][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
        }
        // offset of stage kp + 2: one scalar add where it is needed (as C++ arithmetic the scheduler computes all of a
        // tile's offsets at its top, where they overflow the SGPR file into VGPR lanes)
        if (kp + 2 == NKP) so_run = nx;
        else asm volatile("s_add_u32 %0, %0, 0x800" : "+s"(so_run) : : "scc");
        a_fetch(ar[(kp + PAR) & 1], so_run);
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
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    if (ntw > 0) {
      a_fetch(ar[0], vsw_wave);
      a_fetch(ar[1], vsw_wave + 2048u);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = (0 < NA ? panA : panB)[ni * 64];
      for (int t = 0; t + 1 < ntw; ++t) {
        if (ODD && (t & 1)) tile(t, F_{}, P1{});
        else tile(t, F_{}, P0{});
      }
      if (ODD && ((ntw - 1) & 1)) tile(ntw - 1, T_{}, P1{});
      else tile(ntw - 1, T_{}, P0{});
    } else {                      // a wave without tiles still carries its share of the next panel
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const int ks = wave + 8 * q;
        fetch_ks(nxt, ks < NA ? ks : NA - 1, chunk);
        if (ks < NA) put_ks(nextA + (size_t)ks * SLOT, ks, chunk);
      }
      fetch_B(nxt);
    }
    par ^= 1;
    __syncthreads();     // every wave has finished with the panel (and has stored its part of the next one's A copy)
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

// K -> the instantiated number of k steps (0: none); the kernel pads K up to 4 NKS with zeros
static int hk2_nks(int K) {
  const int ks = (K + 3) / 4;
  if (ks > 52 || ks < 23) return 0;
  // 47..50 steps run in the 52-step instance by default: it needs no scratch (the 50-step one keeps 68 B per lane, 1 GB of
  // HBM traffic per launch at configs[2]) and takes the same time -- at the power-limited clock the MFMAs on the zero padding
  // cost nothing (7.18 / 7.19 ms in the path either way).  hk2_pad52 = 0: the exact-K instance.
  if (ks > 50 || (ks >= 47 && tuning("hk2_pad52", 1))) return 52;
  if (ks > 28) return ks >= 47 ? 50 : 0;
  return ks > 26 ? 28 : 26;
}
static int hk2_na(int nks) { return nks == 50 ? 28 : 26 + (nks == 28 ? 2 : 0); }   // 50 -> 28, 52 -> 26, 28 -> 28, 26 -> 26: (2 NA + NB) <= 78 k steps of 2 KB

static int hk2_lds_limit() {         // per device: the kernels want up to 156 KB of dynamic LDS
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  static int cache[64];
  if (cache[dev]) return cache[dev];
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 0;
  cache[dev] = v;
  return v;
}

bool hk_panel2_applicable(int n0, int n1, int K, long ldh) {
  const int nks = hk2_nks(K);
  if (!nks) return false;
  const int slots = 2 * hk2_na(nks) + (nks - hk2_na(nks));
  return tuning("hk_panel", 1) && tuning("hk_panel2", 1) && n1 >= tuning("hk_panel2_min_n1", 480) &&    // (fewer than 15 tile pairs leave waves without work)
         n0 >= tuning("hk_panel_min_n0", 2048) && ldh <= 7000000L && n1 <= 100000 &&   // (store offsets of a tile pair, (35 ldh + 63) * 8, and the operand's size stay below 2^31)
         hk2_lds_limit() >= slots * 2048;       // (a device with less LDS takes the tiled GEMM that is still in the dispatcher)
}

int hk_panel2_launch(hipStream_t st, const double *d_values, int K, double t, const double *V0, long ld0, int n0,
                     const double *dV1, int ld1, const int *d_idx1, int row0_1, int n1, double *dH, long ldh,
                     double *d_vw) {
  const int nks = hk2_nks(K), nkp = nks / 2, npairs = (n1 + 31) / 32, na = hk2_na(nks);
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
  int dev = 0, n_cu = 0;
  (void)hipGetDevice(&dev);
  {
    static int cu_cache[64];      // keyed by device (the first caller's count was used for every device)
    if (dev >= 0 && dev < 64 && cu_cache[dev]) n_cu = cu_cache[dev];
    else {
      if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
      if (dev >= 0 && dev < 64) cu_cache[dev] = n_cu;
    }
  }
  int grid = n_cu;
  if (grid > g.nblocks) grid = g.nblocks;
  const size_t lds = sizeof(double) * (size_t)(2 * na + (nks - na)) * HP2 * 4;
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
  switch (nks) {
    case 50: if (nt) FLGP_TRY(go(hk_panel2_kernel<50, 28, 2>)); else FLGP_TRY(go(hk_panel2_kernel<50, 28, 0>)); break;
    case 52: if (nt) FLGP_TRY(go(hk_panel2_kernel<52, 26, 2>)); else FLGP_TRY(go(hk_panel2_kernel<52, 26, 0>)); break;
    case 28: if (nt) FLGP_TRY(go(hk_panel2_kernel<28, 28, 2>)); else FLGP_TRY(go(hk_panel2_kernel<28, 28, 0>)); break;
    default: if (nt) FLGP_TRY(go(hk_panel2_kernel<26, 26, 2>)); else FLGP_TRY(go(hk_panel2_kernel<26, 26, 0>)); break;
  }
  return check_launch("hk_panel2_kernel");
}

}  // namespace flgp
