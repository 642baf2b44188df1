// Microbenchmark: peak fp64 VALU FMA rate and fp64 MFMA rate on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ __launch_bounds__(256) void fma_kernel(double *out, double a, double b, int iters) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void mfma_kernel(double *out, double a, double b, int iters) {
  d4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float time_it(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  double *out; hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 4096;
  for (int wpc = 1; wpc <= 8; wpc *= 2) {  // blocks per CU (each block = 4 waves = 1 wave/SIMD)
    int grid = 256 * wpc;
    float ms1 = time_it([&] { hipLaunchKernelGGL(fma_kernel<1>, dim3(grid), dim3(256), 0, 0, out, 0.999, 0.001, iters); }, 5);
    float ms4 = time_it([&] { hipLaunchKernelGGL(fma_kernel<4>, dim3(grid), dim3(256), 0, 0, out, 0.999, 0.001, iters); }, 5);
    float ms8 = time_it([&] { hipLaunchKernelGGL(fma_kernel<8>, dim3(grid), dim3(256), 0, 0, out, 0.999, 0.001, iters); }, 5);
    double fl = 2.0 * grid * 256 * (double)iters;
    printf("VALU f64 FMA  waves/SIMD=%d  1 chain: %.2f TF   4 chains: %.2f TF   8 chains: %.2f TF\n", wpc,
           fl * 1 / ms1 * 1e-9, fl * 4 / ms4 * 1e-9, fl * 8 / ms8 * 1e-9);
  }
  for (int wpc = 1; wpc <= 4; wpc *= 2) {
    int grid = 256 * wpc;
    float m1 = time_it([&] { hipLaunchKernelGGL(mfma_kernel<1>, dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    float m4 = time_it([&] { hipLaunchKernelGGL(mfma_kernel<4>, dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    double fl = 2.0 * 16 * 16 * 4 * (double)grid * 4 * iters;
    printf("MFMA f64 16x16x4  waves/SIMD=%d  1 acc: %.2f TF   4 acc: %.2f TF\n", wpc, fl * 1 / m1 * 1e-9, fl * 4 / m4 * 1e-9);
  }
  return 0;
}
