// What does one extra vector instruction cost the fp64 matrix pipe?  The 77-TF panel loop (B from LDS, A in registers, two waves per
// SIMD) with EX extra instructions of one kind per k step (4 MFMAs): v_add_u32, v_lshl_add_u64, v_add_f64, s_add_u32,
// global_load_dwordx2 (saddr form, L2 hit, result unused), ds_read_b64 (unused).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int HP = 64, NST = 13;
template <int KIND, int EX>
__global__ __launch_bounds__(512, 1) void loop(double *out, const double *rnd, int reps) {
  extern __shared__ double panel[];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
  for (int e = tid; e < NST * 16 * HP; e += 512) panel[e] = rnd[e];
  __syncthreads();
  double a[4];
  for (int kk = 0; kk < 4; ++kk) a[kk] = rnd[(tid * 16 + kk) & 65535];
  d4 acc[4];
  for (int ni = 0; ni < 4; ++ni) acc[ni] = d4{0, 0, 0, 0};
  const double *pan_lane = panel + fr * 4 + fk;
  double bf[4];
  for (int ni = 0; ni < 4; ++ni) bf[ni] = pan_lane[ni * 64];
  unsigned x32 = tid; unsigned long long x64 = tid; double xf = tid; unsigned sx = 0;
  double ld0 = 0, ld1 = 0;
  const unsigned voff = lane * 8u;
  const unsigned ldsoff = (unsigned)(size_t)(fr * 32 + fk * 8);
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
#pragma unroll
        for (int e = 0; e < EX; ++e) {
          if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x32) : "v"(lane));
          if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x64) : "v"(x64));
          if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(xf) : "v"(xf));
          if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
          if (KIND == 5) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ld0) : "v"(voff), "s"(rnd) : "memory");
          if (KIND == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(ld1) : "v"(ldsoff) : "memory");
          if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(x32) : "v"(lane));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bf[ni], acc[ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = bn[ni];
      }
      if (KIND == 5 || KIND == 6) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  }
  double sum = x32 + (double)x64 + xf + sx + ld0 + ld1;
  for (int ni = 0; ni < 4; ++ni) sum += acc[ni][0] + acc[ni][1] + acc[ni][2] + acc[ni][3];
  out[blockIdx.x * 512 + tid] = sum;
}
template <class F> float time_it(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
template <int KIND, int EX> void run(double *out, const double *rnd, const char *what) {
  const int reps = 600, grid = 256;
  const size_t lds = sizeof(double) * NST * 16 * HP;
  (void)hipFuncSetAttribute((const void *)loop<KIND, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float ms = time_it([&] { hipLaunchKernelGGL((loop<KIND, EX>), dim3(grid), dim3(512), lds, 0, out, rnd, reps); });
  const double fl = 2048.0 * 4 * 4 * NST * (double)reps * 8 * grid;
  const double cyc_per_kstep_pair = ms * 1e-3 * 2.4e9 / ((double)reps * NST * 4);   // at 2.4 GHz nominal: 512 = pipe-bound
  printf("%-22s x%d per k step: %.2f ms  %.1f TF   (%.0f nominal cycles per k-step pair; 512 = matrix pipe)\n", what, EX, ms, fl / ms * 1e-9, cyc_per_kstep_pair);
}
int main() {
  double *out, *rnd;
  (void)hipMalloc(&out, sizeof(double) * 1024 * 2048);
  (void)hipMalloc(&rnd, sizeof(double) * 65536);
  double *h = (double *)malloc(sizeof(double) * 65536);
  srand(1);
  for (int i = 0; i < 65536; ++i) { double s = 0; for (int q = 0; q < 12; ++q) s += rand() / (double)RAND_MAX; h[i] = s - 6.0; }
  (void)hipMemcpy(rnd, h, sizeof(double) * 65536, hipMemcpyHostToDevice);
  run<0, 0>(out, rnd, "nothing");
  run<1, 1>(out, rnd, "v_add_u32"); run<1, 4>(out, rnd, "v_add_u32"); run<1, 8>(out, rnd, "v_add_u32");
  run<7, 4>(out, rnd, "v_mov_b32");
  run<2, 1>(out, rnd, "v_lshl_add_u64"); run<2, 4>(out, rnd, "v_lshl_add_u64");
  run<3, 1>(out, rnd, "v_add_f64"); run<3, 4>(out, rnd, "v_add_f64");
  run<4, 4>(out, rnd, "s_add_u32"); run<4, 16>(out, rnd, "s_add_u32");
  run<5, 1>(out, rnd, "global_load_dwordx2"); run<5, 2>(out, rnd, "global_load_dwordx2"); run<5, 4>(out, rnd, "global_load_dwordx2");
  run<6, 1>(out, rnd, "ds_read_b64"); run<6, 4>(out, rnd, "ds_read_b64");
  return 0;
}
