// What a barrier phase costs on gfx950: N rounds of {k dependent LDS reads, one LDS write, __syncthreads()} in one
// workgroup of W waves.  usage: ubench_barrier   (prints cycles per round for W in {1,4,8,16}, k in {0,1,2,3})
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ void kern(int rounds, long long *out, int *sink) {
  __shared__ int buf[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) buf[i] = (i * 7 + 3) & 4095;
  __syncthreads();
  int x = tid & 4095;
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < K; ++k) x = buf[x];          // dependent LDS reads
    buf[(tid * 5 + r) & 4095] = x;                   // one write
    __syncthreads();
  }
  const long long t1 = clock64();
  if (tid == 0) out[0] = t1 - t0;
  if (x == -1) sink[0] = x;
}

int main() {
  long long *d_out; int *d_sink;
  hipMalloc(&d_out, 8); hipMalloc(&d_sink, 4);
  const int rounds = 2000;
  for (int waves : {1, 4, 8, 12, 16}) {
    printf("waves %2d:", waves);
    for (int k = 0; k < 4; ++k) {
      for (int rep = 0; rep < 2; ++rep) {
        if (k == 0) hipLaunchKernelGGL(kern<0>, dim3(1), dim3(64 * waves), 0, 0, rounds, d_out, d_sink);
        if (k == 1) hipLaunchKernelGGL(kern<1>, dim3(1), dim3(64 * waves), 0, 0, rounds, d_out, d_sink);
        if (k == 2) hipLaunchKernelGGL(kern<2>, dim3(1), dim3(64 * waves), 0, 0, rounds, d_out, d_sink);
        if (k == 3) hipLaunchKernelGGL(kern<3>, dim3(1), dim3(64 * waves), 0, 0, rounds, d_out, d_sink);
        hipDeviceSynchronize();
      }
      long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
      printf("  k=%d: %6.0f cyc/round", k, (double)c / rounds);
    }
    printf("\n");
  }
  return 0;
}
