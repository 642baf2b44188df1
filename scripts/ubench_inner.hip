// Microbenchmark: the GEMM inner loop alone (LDS fragment reads + MFMAs on a 64x64 wave tile), no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int GK = 16, GLD = 145, LDB = 130;

template <bool M4, bool SYNC>
__global__ __launch_bounds__(256, 2) void inner(double *out, int stages);

// staged variant: every stage also stores a 128x16 A tile (k-contiguous source pattern: lds[k*LD + row]) and a
// 128x16 B tile (row-contiguous pattern) into the other LDS buffer, optionally fed by global loads
template <int MODE>
__global__ __launch_bounds__(256, 2) void staged(double *out, const double *__restrict__ Ag, const double *__restrict__ Bg,
                                                 int lda, int ldb, int stages) {
  __shared__ double As[2][GK * GLD];
  __shared__ double Bs[2][GK * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64, fr = lane & 15, fk = lane >> 4;
  for (int e = tid; e < GK * GLD; e += 256) { As[0][e] = 1e-3 * e; As[1][e] = 2e-3 * e; Bs[0][e] = 1e-4 * e; Bs[1][e] = 3e-4 * e; }
  __syncthreads();
  d4 acc[4][4];
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0, 0, 0, 0};
  double ra[8], rb[8];
  for (int r = 0; r < 8; ++r) { ra[r] = tid * 1e-5 + r; rb[r] = tid * 2e-5 - r; }
  const int row0 = (blockIdx.x % 32) * 128;
  for (int s = 0; s < stages; ++s) {
    const int cur = s & 1;
    const double *A = As[cur], *B = Bs[cur];
    if (MODE & 1) {
      { const int k = tid & 15, rbk = tid >> 4;     // k-contiguous operand
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) As[cur ^ 1][k * GLD + rbk + 16 * rep] = ra[rep]; }
      { const int row = tid & 127, kb = tid >> 7;   // row-contiguous operand
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) Bs[cur ^ 1][(kb + 2 * rep) * GLD + row] = rb[rep]; }
    }
    if (MODE & 2) {
      const int k0 = (s * 16) % 4096;
      { const int k = k0 + (tid & 15), rbk = tid >> 4;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) ra[rep] = Ag[(size_t)(row0 + rbk + 16 * rep) * lda + k]; }
      { const int row = row0 + (tid & 127), kb = tid >> 7;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) rb[rep] = Bg[(size_t)(k0 + kb + 2 * rep) * ldb + row]; }
    }
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      double fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = A[(kk + fk) * GLD + wr + mi * 16 + fr];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = B[(kk + fk) * GLD + wc + ni * 16 + fr];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }
  double sum = 0;
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  out[blockIdx.x * 256 + tid] = sum;
}


// direct variant: the row-contiguous operand(s) go global -> LDS with global_load_lds_dwordx4 (no VGPR staging,
// no ds_write): wave w fills k rows w, w+4, w+8, w+12 of the other buffer while this stage is multiplied.
// BOTH = 1: both operands that way; BOTH = 0: A through registers (k-contiguous pattern), B direct.
constexpr int GLDD = 146;   // 16-byte aligned rows
template <int BOTH>
__global__ __launch_bounds__(256, 2) void direct(double *out, const double *__restrict__ Ag, const double *__restrict__ Bg,
                                                 int lda, int ldb, int stages) {
  __shared__ __attribute__((aligned(16))) double As[2][GK * GLDD];
  __shared__ __attribute__((aligned(16))) double Bs[2][GK * GLDD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64, fr = lane & 15, fk = lane >> 4;
  for (int e = tid; e < GK * GLDD; e += 256) { As[0][e] = 1e-3 * e; As[1][e] = 2e-3 * e; Bs[0][e] = 1e-4 * e; Bs[1][e] = 3e-4 * e; }
  __syncthreads();
  d4 acc[4][4];
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0, 0, 0, 0};
  double ra[8];
  for (int r = 0; r < 8; ++r) ra[r] = tid * 1e-5 + r;
  const int row0 = (blockIdx.x % 32) * 128;
  for (int s = 0; s < stages; ++s) {
    const int cur = s & 1;
    const double *A = As[cur], *B = Bs[cur];
    const int k0 = ((s + 1) * 16) % 4096;
    // next stage straight into the other buffer
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int k = wave + 4 * rep;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Bg + (size_t)(k0 + k) * ldb + row0 + 2 * lane),
                                       (__attribute__((address_space(3))) void *)&Bs[cur ^ 1][k * GLDD], 16, 0, 0);
      if (BOTH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ag + (size_t)(k0 + k) * lda + row0 + 2 * lane),
                                         (__attribute__((address_space(3))) void *)&As[cur ^ 1][k * GLDD], 16, 0, 0);
    }
    if (!BOTH) {
      { const int k = tid & 15, rbk = tid >> 4;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) As[cur ^ 1][k * GLDD + rbk + 16 * rep] = ra[rep]; }
      { const int k = k0 + (tid & 15), rbk = tid >> 4;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) ra[rep] = Ag[(size_t)(row0 + rbk + 16 * rep) * lda + k]; }
    }
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      double fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = A[(kk + fk) * GLDD + wr + mi * 16 + fr];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = B[(kk + fk) * GLDD + wc + ni * 16 + fr];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's direct loads have landed
    __syncthreads();
  }
  double sum = 0;
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  out[blockIdx.x * 256 + tid] = sum;
}

template <bool M4, bool SYNC>
__global__ __launch_bounds__(256, 2) void inner(double *out, int stages) {
  __shared__ double As[2][GK * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[2][GK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64, fr = lane & 15, fk = lane >> 4;
  for (int e = tid; e < GK * GLD; e += 256) { As[0][e] = 1e-3 * e; As[1][e] = 2e-3 * e; }
  for (int e = tid; e < GK * LDB; e += 256) { Bs[0][e] = 1e-4 * e; Bs[1][e] = 3e-4 * e; }
  __syncthreads();
  d4 acc[4][4];
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = d4{0, 0, 0, 0};
  for (int s = 0; s < stages; ++s) {
    const double *A = As[s & 1], *B = Bs[s & 1];
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      if constexpr (M4) {
        double fa[4]; d2 fb[4][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = A[(kk + fk) * GLD + wr + mi * 16 + fr];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const d2 *pb = (const d2 *)&B[(kk + fk) * LDB + wc + ni * 16 + 4 * (lane & 3)];
          fb[ni][0] = pb[0]; fb[ni][1] = pb[1];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            acc[mi][ni][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[mi], fb[ni][0][0], acc[mi][ni][0], 0, 0, 0);
            acc[mi][ni][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[mi], fb[ni][0][1], acc[mi][ni][1], 0, 0, 0);
            acc[mi][ni][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[mi], fb[ni][1][0], acc[mi][ni][2], 0, 0, 0);
            acc[mi][ni][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[mi], fb[ni][1][1], acc[mi][ni][3], 0, 0, 0);
          }
      } else {
        double fa[4], fb[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = A[(kk + fk) * GLD + wr + mi * 16 + fr];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb[ni] = B[(kk + fk) * LDB + wc + ni * 16 + fr];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (SYNC) __syncthreads();
  }
  double sum = 0;
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  out[blockIdx.x * 256 + tid] = sum;
}

template <class F> float time_it(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
  double *out; (void)hipMalloc(&out, sizeof(double) * 256 * 2048);
  const int stages = 2000;
  for (int grid : {256, 512, 1024}) {
    const double fl = 2.0 * 128 * 128 * 16 * (double)stages * grid;
    float a = time_it([&] { hipLaunchKernelGGL((inner<false, false>), dim3(grid), dim3(256), 0, 0, out, stages); });
    float b = time_it([&] { hipLaunchKernelGGL((inner<false, true>), dim3(grid), dim3(256), 0, 0, out, stages); });
    float c = time_it([&] { hipLaunchKernelGGL((inner<true, false>), dim3(grid), dim3(256), 0, 0, out, stages); });
    float d = time_it([&] { hipLaunchKernelGGL((inner<true, true>), dim3(grid), dim3(256), 0, 0, out, stages); });
    printf("grid %4d: 16x16x4 %.1f TF (barrier %.1f) | 4x4x4 %.1f TF (barrier %.1f)\n", grid, fl / a * 1e-9, fl / b * 1e-9, fl / c * 1e-9, fl / d * 1e-9);
  }
  double *Ag, *Bg;
  (void)hipMalloc(&Ag, sizeof(double) * 4096 * 4096); (void)hipMalloc(&Bg, sizeof(double) * 4096 * 4096);
  (void)hipMemset(Ag, 0, sizeof(double) * 4096 * 4096); (void)hipMemset(Bg, 0, sizeof(double) * 4096 * 4096);
  for (int grid : {512}) {
    const double fl = 2.0 * 128 * 128 * 16 * (double)stages * grid;
    float a = time_it([&] { hipLaunchKernelGGL((staged<0>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    float b = time_it([&] { hipLaunchKernelGGL((staged<1>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    float c = time_it([&] { hipLaunchKernelGGL((staged<2>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    float d = time_it([&] { hipLaunchKernelGGL((staged<3>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    float e = time_it([&] { hipLaunchKernelGGL((direct<0>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    float f = time_it([&] { hipLaunchKernelGGL((direct<1>), dim3(grid), dim3(256), 0, 0, out, Ag, Bg, 4096, 4096, stages); });
    printf("direct-to-LDS grid %d: B direct + A via registers %.1f TF | both direct %.1f TF\n", grid, fl / e * 1e-9, fl / f * 1e-9);
    printf("staged grid %d: compute only %.1f TF | + LDS stores %.1f TF | + global loads %.1f TF | + both %.1f TF\n", grid,
           fl / a * 1e-9, fl / b * 1e-9, fl / c * 1e-9, fl / d * 1e-9);
  }
  return 0;
}
