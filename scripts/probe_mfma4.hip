// Probe: operand/result lane layout of v_mfma_f64_4x4x4_4b_f64 on this device.
// Wave w = (la, lb): A is 1.0 in lane la only, B is 1.0 in lane lb only; the lanes of D that come out
// non-zero tell which (A lane, B lane) pairs meet in which result lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
__global__ void probe(double *out) {
  const int w = blockIdx.x, la = w / 64, lb = w % 64, l = threadIdx.x;
  const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  out[(size_t)w * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main() {
  double *d; hipMalloc(&d, sizeof(double) * 4096 * 64);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  // for each output lane: list of (la, lb)
  for (int o = 0; o < 64; ++o) {
    printf("D lane %2d <-", o);
    for (int w = 0; w < 4096; ++w)
      if (h[(size_t)w * 64 + o] != 0.0) printf(" (a%d,b%d)", w / 64, w % 64);
    printf("\n");
  }
  return 0;
}
