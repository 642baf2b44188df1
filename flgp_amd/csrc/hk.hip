// k7, the contraction itself: H(a, b) = sum_k V0(a, k) Vw(b, k)  with Vw = V[idx1] diag(exp(-t (1 - values)))
// (HK_from_spectrum_cpp, reference src/Spectrum.cpp:90-91) for the shape the path produces: n0 = 1e5 ... 1e7 rows
// of V against a few hundred to a few thousand training rows, K a few hundred.
//
// The tiled GEMM of gemm.hip spends 14 % of such a product filling and draining its pipeline once per 128 x 128
// tile, and the eight tiles of a row block fetch the V rows they share 2.9 times between them (round 2: MFMA busy
// 72 %, 0.59 of the matrix peak).  This kernel turns the loop nest around:
//
//   * one persistent workgroup per CU walks over PANELS of 64 consecutive rows of V0; the panel (64 x K doubles,
//     106 KB at K = 200) is read from HBM exactly once and lives in LDS in the MFMA B-operand order
//     [k / 4][row][k % 4], so that a wave's ds_read_b64 of one fragment covers 512 contiguous bytes (conflict free,
//     no padding);
//   * the small operand Vw (n1 x K, 1.6 MB at m = 1000: L2 resident) is never staged: every wave owns m-tiles of 16
//     rows of Vw and loads its A fragments straight from L2 in the operand layout, four stages (64 k) ahead, in a
//     ring of registers;
//   * nothing in the steady state is shared between waves but the read-only panel: NO barrier inside a panel, the
//     two waves of a SIMD cover each other's waits, and the k loop runs on without a break from one m-tile into the
//     next -- a tile ends with 16 stores of its accumulators and nothing else;
//   * the next panel is fetched into registers while the last m-tile of the current one is multiplied and goes to
//     LDS between two barriers (the only two per panel: ~90 us of MFMAs apart at m = 1000).
//
// This is the GENERAL form (any K <= 288, run-time stage count): MFMA busy 76 %, 7 % faster than the tiled GEMM.  For
// the stage counts of BASELINE's configurations hk2.hip unrolls the k loop and strips the issue stream (86 %).
//
// Arithmetic: acc = v_mfma_f64_16x16x4_f64(a, b, acc) over k ascending from acc = 0 -- the same chain per element
// as gemm_f64_kernel (the instruction adds its four products in ascending k: scripts/probe_mfma_order.hip), so H is
// bit for bit what the tiled GEMM gives (tests/test_gpu_parity.py::test_hk_panel_kernel_bit_identical_to_gemm).
#include "common.h"

namespace flgp {

typedef double hd4 __attribute__((ext_vector_type(4)));
typedef double hd2 __attribute__((ext_vector_type(2)));

constexpr int HP = 64;              // rows of V0 per panel
constexpr int HK_WAVES = 8;         // 512 threads, two waves per SIMD
constexpr int HK_MAX_NST = 18;      // k stages of 16 the panel may hold: 18 * 8 KB = 147 KB of LDS (K <= 288)

struct HkPanelArgs {
  const double *V0; long ld0; int n0;     // V0(a, k) = V0[a + k ld0], a < n0
  const double *Vw; int n1, n1p;          // Vw(b, k) = Vw[b + k n1p]; rows n1 .. n1p and columns K .. 16 nst are zero
  int K, nst;
  double *H; long ldh;                    // H(a, b) = H[a + b ldh]
  int nblocks;                            // panels: ceil(n0 / HP)
};

// PG = k groups (of 4) a thread carries when a panel moves HBM -> registers -> LDS: ceil(4 nst / 8)
template <int PG>
__global__ __launch_bounds__(512, 1) void hk_panel_kernel(HkPanelArgs g) {
  extern __shared__ double panel[];       // [4 nst][HP][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int nst = g.nst, nkg = nst * 4;
  const int nmt = g.n1p >> 4;
  const int ntw = (nmt - wave + HK_WAVES - 1) / HK_WAVES;       // m-tiles of this wave: wave, wave + 8, ...
  const int F = ntw * nst;                                      // stages of 16 k this wave multiplies per panel
  const long n1p = g.n1p;

  // ---- panel transport: wave pg0 takes the k groups pg0, pg0 + 8, ...; lane = row.  Addresses are a wave-uniform
  //      base (SGPRs) plus one 32-bit lane offset, here and below: 28 + 16 + 16 independent 64-bit addresses per
  //      lane do not fit the register file next to the data they fetch.
  const int pj = lane, pg0 = wave;
  double pr[PG][4];
  auto panel_fetch = [&](int blk) {       // unconditional, clamped loads (the padding is zeroed on the way to LDS)
    long a = (long)blk * HP + pj;
    if (a > (long)g.n0 - 1) a = (long)g.n0 - 1;
    const unsigned aoff = (unsigned)(a - (long)blk * HP) * 8u;  // byte offset of the row, clamped at the end of the matrix
    const double *src = g.V0 + (size_t)blk * HP;
#pragma unroll
    for (int q = 0; q < PG; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int k = 4 * (pg0 + 8 * q) + c;
        if (k > g.K - 1) k = g.K - 1;
        const double *col = src + (size_t)k * g.ld0;            // uniform
        pr[q][c] = *(const double *)((const char *)col + aoff);
      }
    }
  };
  auto panel_put = [&]() {
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int kg = pg0 + 8 * q;
      if (kg < nkg) {
        hd2 lo, hi;
        lo[0] = (4 * kg + 0 < g.K) ? pr[q][0] : 0.0;
        lo[1] = (4 * kg + 1 < g.K) ? pr[q][1] : 0.0;
        hi[0] = (4 * kg + 2 < g.K) ? pr[q][2] : 0.0;
        hi[1] = (4 * kg + 3 < g.K) ? pr[q][3] : 0.0;
        hd2 *dst = (hd2 *)(panel + ((size_t)kg * HP + pj) * 4);
        dst[0] = lo;
        dst[1] = hi;
      }
    }
  };

  int blk = blockIdx.x;
  if (blk < g.nblocks) panel_fetch(blk);
  const unsigned vw_lane = (unsigned)(fr + fk * (int)n1p) * 8u; // lane part of an A-fragment address (bytes)
  const double *vw_wave = g.Vw + (size_t)wave * 16;
  const unsigned h_lane = ((unsigned)fr + (unsigned)fk * (unsigned)g.ldh) * 8u;
  const double *pan_lane = panel + (size_t)fr * 4 + fk;

  for (; blk < g.nblocks; blk += gridDim.x) {
    panel_put();
    __syncthreads();
    const int nxt = blk + (int)gridDim.x;
    const bool has_next = nxt < g.nblocks;
    const int pf_at = (F > nst) ? F - nst : 0;     // the stage at which this wave sends for the next panel
    if (F == 0 && has_next) panel_fetch(nxt);

    // ---- A ring: stages f .. f + 3 in flight, loaded unconditionally (past the end: the last stage again)
    double ar[4][4];
    int lt = 0, ls = 0;                            // (m-tile ordinal, stage) of the next load
    auto a_load = [&](double (&dst)[4]) {
      const double *p = vw_wave + (size_t)lt * (16 * HK_WAVES) + (size_t)(ls * 16) * n1p;     // uniform
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dst[kk] = *(const double *)((const char *)(p + (size_t)(4 * kk) * n1p) + vw_lane);
      if (ls + 1 < nst) ++ls;
      else if (lt + 1 < ntw) { ls = 0; ++lt; }
    };
    if (F > 0) {
      a_load(ar[0]); a_load(ar[1]); a_load(ar[2]); a_load(ar[3]);
    }
    hd4 acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = hd4{0.0, 0.0, 0.0, 0.0};
    int ct = 0, cs = 0;                            // (m-tile ordinal, stage) being multiplied
    double bf[4];
    if (F > 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = pan_lane[ni * 64];
    }
    auto stage = [&](double (&a)[4], int f) {
      if (f == pf_at && has_next) panel_fetch(nxt);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // fragments of the next k step (the next stage's first one after kk = 3; harmless re-read at the very end)
        int kgn = cs * 4 + kk + 1;
        if (kgn >= nkg) kgn = 0;
        double bn[4];
        const double *bp = pan_lane + (size_t)kgn * (HP * 4);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bn[ni] = bp[ni * 64];
        // the reads above must be ISSUED before the four MFMAs below: the two waves of a SIMD interleave their MFMAs one
        // by one, so a wave's fourth MFMA issues ~450 cycles after its first and reads issued behind it have ~60 cycles
        // to land before the next k step wants them (left to itself the scheduler puts them there: MFMA busy 76 %)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bf[ni], acc[ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
      }
      if (++cs == nst) {
        // ---- the m-tile is complete: D(row = fk + 4 reg, col = fr) of each 16 x 16 tile
        cs = 0;
        const int bt = (wave + HK_WAVES * ct) * 16;            // uniform
        ++ct;
        double *hb = g.H + (size_t)blk * HP + (size_t)bt * g.ldh;   // uniform
        const long a0 = (long)blk * HP + fr;
        const int b0 = bt + fk;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            if (a0 + ni * 16 < (long)g.n0 && b0 + 4 * reg < g.n1)
              *(double *)((char *)(hb + ni * 16 + (size_t)(4 * reg) * g.ldh) + h_lane) = acc[ni][reg];
          }
          acc[ni] = hd4{0.0, 0.0, 0.0, 0.0};
        }
      }
    };
    for (int f = 0; f < F; f += 4) {
      stage(ar[0], f);
      a_load(ar[0]);
      if (f + 1 < F) stage(ar[1], f + 1);
      a_load(ar[1]);
      if (f + 2 < F) stage(ar[2], f + 2);
      a_load(ar[2]);
      if (f + 3 < F) stage(ar[3], f + 3);
      a_load(ar[3]);
    }
    __syncthreads();     // every wave has finished with the panel
  }
}

// Vw(b, k) = exp(-t (1 - values_k)) * V1(row(b), k) for b < n1, k < K, zero in the padding (b < n1p, k < Kp)
__global__ void hk_scale_pad_kernel(const double *__restrict__ values, int K, int Kp, double t,
                                    const double *__restrict__ V1, int ld1, const int *__restrict__ idx1, int row0,
                                    int n1, int n1p, double *__restrict__ Vw) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)n1p * Kp) return;
  const int b = (int)(e % n1p), k = (int)(e / n1p);
  double v = 0.0;
  if (b < n1 && k < K) {
    const int row = idx1 ? idx1[b] : row0 + b;
    const double w = exp(-t * (1.0 - values[k]));  // src/Spectrum.cpp:86,90
    v = V1[(size_t)k * ld1 + row] * w;
  }
  Vw[e] = v;
}

// per-device figures (the first caller's CU count was used for every device, and a device with less LDS than
// the panel needs made the launch fail instead of taking the tiled GEMM that is still in the dispatcher)
static void hk_device_figures(int *n_cu, int *lds_max) {
  static int cu_cache[64], lds_cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cu_cache[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cu_cache[dev] = v;
    v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 65536;
    lds_cache[dev] = v;
  }
  if (n_cu) *n_cu = cu_cache[dev];
  if (lds_max) *lds_max = lds_cache[dev];
}

bool hk_panel_applicable(int n0, int n1, int K, long ldh) {
  const int nst = (K + 15) / 16;
  int lds_max = 0;
  hk_device_figures(nullptr, &lds_max);
  return tuning("hk_panel", 1) && nst <= HK_MAX_NST && n1 >= tuning("hk_panel_min_n1", 64) &&
         n0 >= tuning("hk_panel_min_n0", 2048) && ldh <= 150000000L &&   // (a lane's store offset (3 ldh + 15) * 8 stays in 32 bits)
         (size_t)lds_max >= sizeof(double) * (size_t)nst * 16 * HP;      // (the panel in LDS)
}

// doubles of the small operand's buffer: the larger of the two layouts (hk.hip: rows padded to 16; hk2.hip: to tile pairs)
size_t hk_panel_vw_elems(int n1, int K) {
  const size_t n1p = (size_t)(n1 + 31) / 32 * 32, Kp = (size_t)(K + 15) / 16 * 16;
  return n1p * Kp;
}

// H(a, b) = sum_k V0(a, k) Vw(b, k) with Vw built from (values, V1[rows]) into d_vw (hk_panel_vw_elems doubles)
int hk_panel_launch(hipStream_t st, const double *d_values, int K, double t, const double *V0, long ld0, int n0,
                    const double *dV1, int ld1, const int *d_idx1, int row0_1, int n1, double *dH, long ldh,
                    double *d_vw) {
  const int nst = (K + 15) / 16, Kp = nst * 16, n1p = (n1 + 15) / 16 * 16;
  hipLaunchKernelGGL(hk_scale_pad_kernel, dim3(ceil_div((long)n1p * Kp, 256)), dim3(256), 0, st, d_values, K, Kp, t, dV1,
                     ld1, d_idx1, row0_1, n1, n1p, d_vw);
  FLGP_TRY(check_launch("hk_scale_pad_kernel"));
  HkPanelArgs g;
  g.V0 = V0; g.ld0 = ld0; g.n0 = n0;
  g.Vw = d_vw; g.n1 = n1; g.n1p = n1p;
  g.K = K; g.nst = nst;
  g.H = dH; g.ldh = ldh;
  g.nblocks = ceil_div(n0, HP);
  int n_cu = 256;
  hk_device_figures(&n_cu, nullptr);
  int grid = n_cu * tuning("hk_panel_wg_per_cu", 1);
  if (grid > g.nblocks) grid = g.nblocks;
  const size_t lds = sizeof(double) * (size_t)nst * 16 * HP;
  const double fl = 2.0 * (double)n0 * (double)n1 * (double)K;
  ProfScope ps("hk_panel_kernel", st, fl);
  auto go = [&](auto kfn) -> int {
    FLGP_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, g);
    return FLGP_OK;
  };
  if (nst <= 14) FLGP_TRY(go(hk_panel_kernel<7>));
  else FLGP_TRY(go(hk_panel_kernel<9>));
  return check_launch("hk_panel_kernel");
}

}  // namespace flgp
