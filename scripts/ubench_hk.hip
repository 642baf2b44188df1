// Microbenchmark for the panel contraction's inner loop: B fragments from an LDS-resident panel ([k/4][row][k%4], 64 rows),
// A fragments held in registers (no global traffic in the loop), MT m-tiles of 16 rows per wave, WPS waves per SIMD.
// Random operands (the chip's clock under MFMA load depends on the data).  Prints TFLOP/s per configuration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int HP = 64, NST = 13;

template <int MT, int NT, bool SCHED>
__global__ __launch_bounds__(NT, 1) void loop(double *out, const double *rnd, int reps) {
  extern __shared__ double panel[];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
  for (int e = tid; e < NST * 16 * HP; e += NT) panel[e] = rnd[e];
  __syncthreads();
  double a[MT][4];
  for (int m = 0; m < MT; ++m) for (int kk = 0; kk < 4; ++kk) a[m][kk] = rnd[(tid * 16 + m * 4 + kk) & 65535];
  d4 acc[MT][4];
  for (int m = 0; m < MT; ++m) for (int ni = 0; ni < 4; ++ni) acc[m][ni] = d4{0, 0, 0, 0};
  const double *pan_lane = panel + fr * 4 + fk;
  double bf[4];
  for (int ni = 0; ni < 4; ++ni) bf[ni] = pan_lane[ni * 64];
  for (int r = 0; r < reps; ++r) {
    for (int st = 0; st < NST; ++st) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        int kgn = st * 4 + kk + 1;
        if (kgn >= NST * 4) kgn = 0;
        double bn[4];
        const double *bp = pan_lane + kgn * (HP * 4);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bn[ni] = bp[ni * 64];
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[m][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][kk], bf[ni], acc[m][ni], 0, 0, 0);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
      }
    }
  }
  double sum = 0;
  for (int m = 0; m < MT; ++m) for (int ni = 0; ni < 4; ++ni) sum += acc[m][ni][0] + acc[m][ni][1] + acc[m][ni][2] + acc[m][ni][3];
  out[blockIdx.x * NT + tid] = sum;
}

// no LDS at all: both operands in registers (the matrix pipe's own ceiling at the clock the chip holds)
template <int MT, int NT>
__global__ __launch_bounds__(NT, 1) void bare(double *out, const double *rnd, int reps) {
  const int tid = threadIdx.x;
  double a[MT][4], b[4][4];
  for (int m = 0; m < MT; ++m) for (int kk = 0; kk < 4; ++kk) a[m][kk] = rnd[(tid * 16 + m * 4 + kk) & 65535];
  for (int ni = 0; ni < 4; ++ni) for (int kk = 0; kk < 4; ++kk) b[ni][kk] = rnd[(tid * 16 + ni * 4 + kk + 7777) & 65535];
  d4 acc[MT][4];
  for (int m = 0; m < MT; ++m) for (int ni = 0; ni < 4; ++ni) acc[m][ni] = d4{0, 0, 0, 0};
  for (int r = 0; r < reps * NST; ++r) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[m][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][kk], b[ni][kk], acc[m][ni], 0, 0, 0);
  }
  double sum = 0;
  for (int m = 0; m < MT; ++m) for (int ni = 0; ni < 4; ++ni) sum += acc[m][ni][0] + acc[m][ni][1] + acc[m][ni][2] + acc[m][ni][3];
  out[blockIdx.x * NT + tid] = sum;
}

template <class F> float time_it(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}


// production-like loop with toggles: LOADS = A fragments from an L2-resident matrix through a 4-stage register ring,
// STORES = 16 stores at the end of every m-tile (13 stages), both exactly as in csrc/hk.hip
template <int LOADS, bool STORES>
__global__ __launch_bounds__(512, 1) void prodlike(double *out, const double *rnd, const double *Vw, int n1p, int ntw, double *H, long ldh) {
  extern __shared__ double panel[];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < NST * 16 * HP; e += 512) panel[e] = rnd[e];
  __syncthreads();
  const int nst = NST, nkg = NST * 4;
  const int F = ntw * nst;
  const unsigned vw_lane = (unsigned)(fr + fk * n1p) * 8u;
  const double *vw_wave = Vw + (size_t)wave * 16;
  const unsigned h_lane = ((unsigned)fr + (unsigned)fk * (unsigned)ldh) * 8u;
  const double *pan_lane = panel + fr * 4 + fk;
  double ar[4][4];
  int lt = 0, ls = 0;
  double dump[4] = {0, 0, 0, 0};
  auto a_load = [&](double (&dst)[4]) {
    if constexpr (LOADS == 2) {
      // swizzled operand: [tile][stage][lane][4] -- a lane's four k steps of a stage are 32 contiguous bytes
      typedef double dd2 __attribute__((ext_vector_type(2)));
      const double *p = Vw + ((size_t)(wave + 8 * (lt & 7)) * nst + ls) * 256;      // uniform
      const dd2 lo = *(const dd2 *)((const char *)p + lane * 32u), hi = *(const dd2 *)((const char *)p + lane * 32u + 16u);
      dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = hi[0]; dst[3] = hi[1];
    } else if constexpr (LOADS == 3) {
      const double *p = vw_wave + (size_t)(lt & 7) * 128 + (size_t)(ls * 16) * n1p;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dump[kk] += *(const double *)((const char *)(p + (size_t)(4 * kk) * n1p) + vw_lane);
    } else {
      const double *p = vw_wave + (size_t)(lt & 7) * 128 + (size_t)(ls * 16) * n1p;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dst[kk] = *(const double *)((const char *)(p + (size_t)(4 * kk) * n1p) + vw_lane);
    }
    if (ls + 1 < nst) ++ls;
    else if (lt + 1 < ntw) { ls = 0; ++lt; }
  };
  a_load(ar[0]); a_load(ar[1]); a_load(ar[2]); a_load(ar[3]);
  d4 acc[4];
  for (int ni = 0; ni < 4; ++ni) acc[ni] = d4{0, 0, 0, 0};
  int ct = 0, cs = 0;
  double bf[4];
  for (int ni = 0; ni < 4; ++ni) bf[ni] = pan_lane[ni * 64];
  const int blk = blockIdx.x;
  auto stage = [&](double (&a)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int kgn = cs * 4 + kk + 1;
      if (kgn >= nkg) kgn = 0;
      double bn[4];
      const double *bp = pan_lane + (size_t)kgn * (HP * 4);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bn[ni] = bp[ni * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bf[ni], acc[ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
    }
    if (++cs == nst) {
      cs = 0;
      const int bt = (wave + 8 * (ct & 7)) * 16;
      ++ct;
      if (STORES) {
        double *hb = H + (size_t)blk * HP + (size_t)bt * ldh;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) *(double *)((char *)(hb + ni * 16 + (size_t)(4 * reg) * ldh) + h_lane) = acc[ni][reg];
          acc[ni] = d4{0, 0, 0, 0};
        }
      }
    }
  };
  for (int f = 0; f < F; f += 4) {
    stage(ar[0]); if (LOADS) a_load(ar[0]);
    if (f + 1 < F) stage(ar[1]);
    if (LOADS) a_load(ar[1]);
    if (f + 2 < F) stage(ar[2]);
    if (LOADS) a_load(ar[2]);
    if (f + 3 < F) stage(ar[3]);
    if (LOADS) a_load(ar[3]);
  }
  double sum = 0;
  for (int ni = 0; ni < 4; ++ni) sum += acc[ni][0] + acc[ni][1] + acc[ni][2] + acc[ni][3];
  out[blockIdx.x * 512 + tid] = sum + dump[0] + dump[1] + dump[2] + dump[3];
}
template <int LOADS, bool STORES> void run_prod(double *out, const double *rnd, const double *Vw, double *H) {
  const int grid = 256, n1p = 1008, ntw = 512;      // 64 m-tiles per wave (8 panels' worth), rows wrap inside Vw by construction
  const size_t lds = sizeof(double) * NST * 16 * HP;
  (void)hipFuncSetAttribute((const void *)prodlike<LOADS, STORES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float ms = time_it([&] { hipLaunchKernelGGL((prodlike<LOADS, STORES>), dim3(grid), dim3(512), lds, 0, out, rnd, Vw, n1p, ntw, H, 1000000L); });
  const double fl = 2048.0 * 4 * 4 * NST * (double)ntw * 8 * grid;
  printf("production-like loop, loads=%d stores=%d: %.2f ms  %.1f TF\n", (int)LOADS, (int)STORES, ms, fl / ms * 1e-9);
}

template <int MT, int NT, bool SCHED> void run(double *out, const double *rnd, const char *tag) {
  const int reps = 2000 / MT, grid = 256;
  const size_t lds = sizeof(double) * NST * 16 * HP;
  (void)hipFuncSetAttribute((const void *)loop<MT, NT, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float ms = time_it([&] { hipLaunchKernelGGL((loop<MT, NT, SCHED>), dim3(grid), dim3(NT), lds, 0, out, rnd, reps); });
  const double fl = 2048.0 * MT * 4 * 4 * NST * (double)reps * (NT / 64) * grid;
  printf("%-44s MT=%d waves/SIMD=%d sched=%d: %.2f ms  %.1f TF\n", tag, MT, NT / 256, (int)SCHED, ms, fl / ms * 1e-9);
}
template <int MT, int NT> void run_bare(double *out, const double *rnd) {
  const int reps = 2000 / MT, grid = 256;
  float ms = time_it([&] { hipLaunchKernelGGL((bare<MT, NT>), dim3(grid), dim3(NT), 0, 0, out, rnd, reps); });
  const double fl = 2048.0 * MT * 4 * 4 * NST * (double)reps * (NT / 64) * grid;
  printf("%-44s MT=%d waves/SIMD=%d        : %.2f ms  %.1f TF\n", "bare (operands in registers)", MT, NT / 256, ms, fl / ms * 1e-9);
}

int main() {
  double *out, *rnd;
  (void)hipMalloc(&out, sizeof(double) * 1024 * 2048);
  (void)hipMalloc(&rnd, sizeof(double) * 65536);
  double *h = (double *)malloc(sizeof(double) * 65536);
  srand(1);
  for (int i = 0; i < 65536; ++i) {   // ~N(0,1): sum of 12 uniforms
    double s = 0; for (int q = 0; q < 12; ++q) s += rand() / (double)RAND_MAX; h[i] = s - 6.0;
  }
  (void)hipMemcpy(rnd, h, sizeof(double) * 65536, hipMemcpyHostToDevice);
  run_bare<1, 512>(out, rnd);
  run<1, 512, true>(out, rnd, "panel loop"); run<2, 512, true>(out, rnd, "panel loop");
  {
    // Vw: (8 waves x 64 tiles x 16 rows) x 208 columns would be 8192 rows; the loop only advances lt to ntw - 1 = 63, so rows < 8 * 16 + 63 * 128 + 16
    const size_t rows = 8 * 16 + 64 * 128 + 16, cols = 208;
    double *Vw, *H;
    (void)hipMalloc(&Vw, sizeof(double) * 1008 * (cols + 8 * rows / 1008 + 16));
    (void)hipMalloc(&H, sizeof(double) * 1000000L * 1040);
    (void)hipMemset(Vw, 0, sizeof(double) * 1008 * (cols + 8 * rows / 1008 + 16));
    for (size_t o = 0; o + 65536 <= 1008 * (cols + 8 * rows / 1008 + 16); o += 65536) (void)hipMemcpy(Vw + o, rnd, sizeof(double) * 65536, hipMemcpyDeviceToDevice);
    run_prod<0, false>(out, rnd, Vw, H); run_prod<1, false>(out, rnd, Vw, H); run_prod<2, false>(out, rnd, Vw, H); run_prod<3, false>(out, rnd, Vw, H); run_prod<0, true>(out, rnd, Vw, H); run_prod<1, true>(out, rnd, Vw, H); run_prod<2, true>(out, rnd, Vw, H);
  }
  return 0;
}
