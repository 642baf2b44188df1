// Probe: is v_mfma_f64_16x16x4_f64 (and the 4x4x4 form) a chain of IEEE FMAs, and in which k order?
// D(i,j) from the instruction is compared bit for bit with fma(a_p3,b_p3, fma(a_p2,b_p2, fma(a_p1,b_p1, fma(a_p0,b_p0, c))))
// for all 24 orders p, on random operands (wide exponent range to provoke rounding differences).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void run16(const double *A, const double *B, const double *C, double *D, int trials) {
  const int l = threadIdx.x;
  for (int t = blockIdx.x; t < trials; t += gridDim.x) {
    const double a = A[(size_t)t * 64 + l], b = B[(size_t)t * 64 + l];   // A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(size_t)t * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
  }
}
static uint64_t s = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double rd(int spread) {
  double m = (double)(rnd() >> 11) * 0x1p-53 + 0.5;
  int e = (int)(rnd() % (2 * spread + 1)) - spread;
  return ((rnd() & 1) ? -m : m) * ldexp(1.0, e);
}
int main() {
  const int T = 20000;
  std::vector<double> A(T * 64), B(T * 64), C(T * 256), D(T * 256);
  for (auto &v : A) v = rd(8);
  for (auto &v : B) v = rd(8);
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < 256; ++e) C[(size_t)t * 256 + e] = (t % 3 == 0) ? 0.0 : rd(10);
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run16, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, T);
  hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
  int perm[4] = {0, 1, 2, 3};
  do {
    long bad = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double acc = C[(size_t)t * 256 + i * 16 + j];
          for (int q = 0; q < 4; ++q) {
            const int k = perm[q];
            acc = fma(A[(size_t)t * 64 + i + 16 * k], B[(size_t)t * 64 + j + 16 * k], acc);
          }
          const double d = D[(size_t)t * 256 + i * 16 + j];
          if (memcmp(&d, &acc, 8) != 0) ++bad;
        }
    printf("order %d%d%d%d: %ld mismatches of %ld\n", perm[0], perm[1], perm[2], perm[3], bad, (long)T * 256);
  } while (std::next_permutation(perm, perm + 4));
  return 0;
}
