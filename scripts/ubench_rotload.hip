// Microbenchmark: only the OPERAND TRAFFIC of the solver's rotation (5000 x 256)(256 x 256) as the 64-tile GEMM issues it --
// per workgroup and 16-deep stage 64 rows x 16 k of X (row-contiguous, 16-byte loads) and 64 columns x 16 k of W
// (k-contiguous) -- no LDS, no MFMA: is the memory system what a stage waits for?
// usage: ubench_rotload [rows] [depth: loads of how many stages in flight]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void rotload(const double *__restrict__ X, const double *__restrict__ W, int s, int b,
                                                   double *__restrict__ out, int stages) {
  const int tid = threadIdx.x;
  const int ntm = b / 64;
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;      // tm: column tile of W (fastest), tn: row tile of X
  int row0 = tn * 64; if (row0 + 64 > s) row0 = s - 64;
  const int col0 = tm * 64;
  // RC: rows 2l, 2l+1 at k = tid/32 + 8 rep;   KC: k pair 2 (tid & 7) of column tid >> 3 + 32 rep
  const double *px = X + row0 + 2 * (tid & 31) + (size_t)(tid >> 5) * s;
  const double *pw = W + (size_t)(col0 + (tid >> 3)) * b + 2 * (tid & 7);
  d2 acc = {0.0, 0.0};
  d2 rx[DEPTH][2], rw[DEPTH][2];
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) {
    const int k0 = (q < stages ? q : stages - 1) * 16;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      rx[q][rep] = *(const d2 *)(px + (size_t)(k0 + 8 * rep) * s);
      rw[q][rep] = *(const d2 *)(pw + (size_t)(32 * rep) * b + k0);
    }
  }
  for (int st = 0; st < stages; st += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      acc += rx[q][0] + rx[q][1] + rw[q][0] + rw[q][1];
      const int sn = st + q + DEPTH;
      const int k0 = (sn < stages ? sn : stages - 1) * 16;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        rx[q][rep] = *(const d2 *)(px + (size_t)(k0 + 8 * rep) * s);
        rw[q][rep] = *(const d2 *)(pw + (size_t)(32 * rep) * b + k0);
      }
    }
  }
  if (acc[0] + acc[1] == 1.2345e300) out[blockIdx.x * 256 + tid] = acc[0];
}

int main(int argc, char **argv) {
  const int b = 256, stages = 16;
  double *W, *out;
  hipMalloc(&W, sizeof(double) * b * b); hipMalloc(&out, sizeof(double) * 4096 * 256);
  hipMemset(W, 0, sizeof(double) * b * b);
  for (int s : {640, 1280, 2560, 5000, 10000, 20000}) {
    double *X; hipMalloc(&X, sizeof(double) * (size_t)s * b); hipMemset(X, 0, sizeof(double) * (size_t)s * b);
    const int nwg = (s + 63) / 64 * (b / 64);
    for (int depth : {1, 2, 4}) {
      auto run = [&]() {
        if (depth == 1) hipLaunchKernelGGL(rotload<1>, dim3(nwg), dim3(256), 0, 0, X, W, s, b, out, stages);
        else if (depth == 2) hipLaunchKernelGGL(rotload<2>, dim3(nwg), dim3(256), 0, 0, X, W, s, b, out, stages);
        else hipLaunchKernelGGL(rotload<4>, dim3(nwg), dim3(256), 0, 0, X, W, s, b, out, stages);
      };
      for (int i = 0; i < 5; ++i) run();
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 50; ++i) run();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("rows %6d  workgroups %5d  depth %d: %7.2f us per launch  (%.2f us per stage, %.2f TB/s of operand bytes)\n", s, nwg, depth,
             ms / 50 * 1e3, ms / 50 * 1e3 / stages, (double)nwg * stages * 16384 / (ms / 50 * 1e-3) / 1e12);
    }
    hipFree(X);
  }
  return 0;
}
