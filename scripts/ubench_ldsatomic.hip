// Microbenchmark: throughput of ds_add_f64 (LDS f64 atomic add, no return) by number of active lanes and waves per CU,
// and the order in which lanes of ONE instruction that hit the same address are applied.
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/bin/ubench_ldsatomic scripts/ubench_ldsatomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__global__ __launch_bounds__(64) void k_atomic(int active, int iters, int stride, double *out, long long *cyc) {
  extern __shared__ double tab[];
  const int lane = threadIdx.x;
  for (int j = lane; j < 5000; j += 64) tab[j] = 0.0;
  __syncthreads();
  const int addr0 = (lane * stride) % 5000;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (lane < active)
      __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&tab[(addr0 + it * 7) % 5000], 1.0 + lane);
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  double s = 0.0;
  for (int j = lane; j < 5000; j += 64) s += tab[j];
  out[blockIdx.x * 64 + lane] = s;
}

// all 64 lanes add to ONE address values of very different magnitude: the result tells the order of application
__global__ __launch_bounds__(64) void k_order(const double *vals, double *out) {
  __shared__ double cell[2];
  if (threadIdx.x == 0) { cell[0] = 0.0; cell[1] = 0.0; }
  __syncthreads();
  __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&cell[0], vals[threadIdx.x]);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = cell[0];
}

int main() {
  int ncu = 256;
  double *out; long long *cyc;
  hipMalloc(&out, sizeof(double) * 64 * 4096); hipMalloc(&cyc, sizeof(long long) * 4096);
  const int iters = 4000;
  for (int wpc : {1, 2, 4}) {
    for (int active : {1, 10, 20, 40, 60, 64}) {
      for (int stride : {1, 13}) {
        const int grid = ncu * wpc;
        hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(64), 40000, 0, active, iters, stride, out, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(64), 40000, 0, active, iters, stride, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= grid;
        printf("waves/CU %d active %2d stride %2d: %.1f memtime-ticks per instr per wave, kernel %.3f ms -> %.1f ns per instr per CU\n", wpc, active, stride,
               avg / iters, ms, ms * 1e6 / iters / wpc);
      }
    }
  }
  // order probe
  std::vector<double> v(64);
  for (int l = 0; l < 64; ++l) v[l] = std::ldexp(1.0 + l * 0.01, (l * 7) % 60 - 30);
  double seq = 0.0; for (int l = 0; l < 64; ++l) seq += v[l];
  double rev = 0.0; for (int l = 63; l >= 0; --l) rev += v[l];
  double *dv; hipMalloc(&dv, sizeof(double) * 64); hipMemcpy(dv, v.data(), sizeof(double) * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_order, dim3(8), dim3(64), 0, 0, dv, out);
  double r[8]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  printf("order probe: device %.17g (x8 equal: %d), ascending-lane %.17g, descending-lane %.17g\n", r[0],
         (int)(r[0] == r[1] && r[1] == r[7]), seq, rev);
  return 0;
}
