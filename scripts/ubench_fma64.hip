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

// MFMA and VALU FMA interleaved in one wave: NM MFMAs (independent accumulators) + NV FMAs per iteration
template <int NM, int NV>
__global__ __launch_bounds__(256) void mixed_kernel(double *out, double a, double b, int iters) {
  d4 acc[NM];
  double va[NV > 0 ? NV : 1];
#pragma unroll
  for (int c = 0; c < NM; ++c) acc[c] = d4{0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < NV; ++c) va[c] = threadIdx.x * 1e-3 + c;
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < NM; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
#pragma unroll
      for (int v = c * NV / NM; v < (c + 1) * NV / NM; ++v) va[v] = __builtin_fma(va[v], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < NM; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
  for (int c = 0; c < NV; ++c) s += va[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
__global__ __launch_bounds__(256) void mfma4_kernel(double *out, double a, double b, int iters) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0;
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
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
  for (int wpc = 1; wpc <= 2; wpc *= 2) {
    int grid = 256 * wpc;
    double flm = 2.0 * 16 * 16 * 4 * (double)grid * 4 * iters * 4, flv1 = 2.0 * grid * 256 * (double)iters;
    float t0 = time_it([&] { hipLaunchKernelGGL((mixed_kernel<4, 0>), dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    float t8 = time_it([&] { hipLaunchKernelGGL((mixed_kernel<4, 8>), dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    float t16 = time_it([&] { hipLaunchKernelGGL((mixed_kernel<4, 16>), dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    float t32 = time_it([&] { hipLaunchKernelGGL((mixed_kernel<4, 32>), dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    printf("mixed waves/SIMD=%d  4 MFMA: %.2f TF | +8 FMA: %.2f TF | +16 FMA: %.2f TF | +32 FMA: %.2f TF (MFMA+VALU total)\n", wpc,
           flm / t0 * 1e-9, (flm + 8 * flv1) / t8 * 1e-9, (flm + 16 * flv1) / t16 * 1e-9, (flm + 32 * flv1) / t32 * 1e-9);
    float q = time_it([&] { hipLaunchKernelGGL(mfma4_kernel<8>, dim3(grid), dim3(256), 0, 0, out, 0.5, 0.25, iters); }, 5);
    printf("MFMA f64 4x4x4 (4 blocks) waves/SIMD=%d  8 acc: %.2f TF\n", wpc, 2.0 * 4 * 4 * 4 * 4 * (double)grid * 4 * iters * 8 / q * 1e-9);
  }
  // short vs long runs (clock behaviour under sustained fp64 MFMA load)
  for (int it2 : {256, 4096, 65536}) {
    float m4 = time_it([&] { hipLaunchKernelGGL(mfma_kernel<4>, dim3(512), dim3(256), 0, 0, out, 0.5, 0.25, it2); }, 3);
    printf("MFMA f64 16x16x4 iters=%d: %.3f ms  %.2f TF\n", it2, m4, 2.0 * 16 * 16 * 4 * 512.0 * 4 * it2 * 4 / m4 * 1e-9);
  }
  return 0;
}
