// SURVEY 8f-4, second half: mini-batch k-means anchors on the device.
//
// subsample_cpp(method = "minibatchkmeans") (reference src/Utils.cpp:49-62) calls back into R:
//   ClusterR::MiniBatchKmeans(data = X, clusters = s, batch_size = 10 s, init_fraction = 20 s / n, num_init = nstart)
// -- every other argument at ClusterR's default (max_iters = 100, initializer = "kmeans++", early_stop_iter = 10) -- and
// then counts 1-NN assignments with KNN_cpp(X, centres, 1).  ClusterR is a third-party R package (version unpinned
// in DESCRIPTION; not under /root/reference) that draws from R's RNG, so its centres cannot be reproduced value for value
// outside R.  What is built here is the published algorithm it implements, with exactly the reference's parameters, on
// this library's own seeded counter RNG:
//   * k-means++ seeding (Arthur & Vassilvitskii 2007) on a random subsample of init_fraction * n rows: every next centre
//     is drawn with probability proportional to the squared distance to the nearest centre chosen so far;
//   * mini-batch updates (Sculley 2010): per iteration batch_size distinct rows, each assigned to its nearest centre as
//     the centres stood at the start of the iteration (the k-NN kernel with r = 1 on the gathered batch), then per centre,
//     in batch order,  v_c += 1,  c <- (1 - 1/v_c) c + (1/v_c) x  (updates of different centres commute);
//   * early stop: when the batch's sum of squared distances has not improved for early_stop_iter iterations in a row;
//   * num_init starts, the one with the smallest total within-cluster sum of squares over ALL points wins;
//   * sizes = 1-NN counts of all points (the k-NN kernel with r = 1: the arithmetic of src/Utils.cpp:59-62).
// Every floating-point operation has a fixed order (below), so the CPU restatement oracle.np_kmeans_minibatch agrees bit
// for bit; against ClusterR itself agreement can only be in distribution (tested against Lloyd's within-SS).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace flgp {

// squared distance, coordinates ascending, one rounded multiply and one rounded add per coordinate
__device__ __forceinline__ double mb_dist2(const double *__restrict__ X, int ldx, long row, const double *__restrict__ c, int d) {
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    const double df = X[(size_t)k * ldx + row] - c[k];
    acc = acc + df * df;
  }
  return acc;
}

constexpr int MB_BLK = 1024;

// d2[i] = min(d2[i], |x_sub[i] - cnew|^2) (first == 1: no min); bsum[b] = d2[b*1024] + d2[b*1024+1] + ... left to right
__global__ __launch_bounds__(MB_BLK) void mb_pp_update_kernel(const double *__restrict__ X, int ldx, int d, const int *__restrict__ sub,
                                                              int nsub, const double *__restrict__ cnew, int first,
                                                              double *__restrict__ d2, double *__restrict__ bsum) {
  const int i = blockIdx.x * MB_BLK + threadIdx.x;
  if (i < nsub) {
    const double v = mb_dist2(X, ldx, sub[i], cnew, d);
    d2[i] = (first || v < d2[i]) ? v : d2[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int i0 = blockIdx.x * MB_BLK, i1 = min(i0 + MB_BLK, nsub);
    double acc = 0.0;
    for (int q = i0; q < i1; ++q) acc = acc + d2[q];
    bsum[blockIdx.x] = acc;
  }
}

// The next centre: target = u * total (total = the block sums added left to right); the first element whose running
// sum -- block sums first, then inside the block, all left to right -- exceeds target.  Copies its row into centre c of C
// (s x d column-major) and into cnew.
__global__ void mb_pp_pick_kernel(const double *__restrict__ X, int ldx, int d, const int *__restrict__ sub, int nsub,
                                  const double *__restrict__ d2, const double *__restrict__ bsum, int nblk, double u,
                                  double *__restrict__ C, int s, int c, double *__restrict__ cnew, int *__restrict__ picked) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0;
  for (int b = 0; b < nblk; ++b) total = total + bsum[b];
  const double target = u * total;
  double acc = 0.0;
  int blk = -1;
  for (int b = 0; b < nblk; ++b) {
    const double nx = acc + bsum[b];
    if (nx > target) { blk = b; break; }
    acc = nx;
  }
  if (blk < 0) {        // (u * total rounded up to total: the last block, scanned from its own running sum)
    blk = nblk - 1;
    acc = 0.0;
    for (int b = 0; b < blk; ++b) acc = acc + bsum[b];
  }
  const int i0 = blk * MB_BLK, i1 = min(i0 + MB_BLK, nsub);
  int pick = i1 - 1;
  for (int q = i0; q < i1; ++q) {
    acc = acc + d2[q];
    if (acc > target) { pick = q; break; }
  }
  const long row = sub[pick];
  for (int k = 0; k < d; ++k) {
    const double v = X[(size_t)k * ldx + row];
    C[(size_t)k * s + c] = v;
    cnew[k] = v;
  }
  picked[c] = pick;
}

// first centre: row sub[pick]
__global__ void mb_pp_first_kernel(const double *__restrict__ X, int ldx, int d, const int *__restrict__ sub, int pick,
                                   double *__restrict__ C, int s, double *__restrict__ cnew, int *__restrict__ picked) {
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    const double v = X[(size_t)k * ldx + sub[pick]];
    C[(size_t)k * s] = v;
    cnew[k] = v;
  }
  if (threadIdx.x == 0) picked[0] = pick;
}

// Xb(p, k) = X(batch[p], k): the batch as a B x d column-major block
__global__ void mb_gather_kernel(const double *__restrict__ X, int ldx, int d, const int *__restrict__ batch, int B,
                                 double *__restrict__ Xb, const int *__restrict__ state) {
  if (state[2]) return;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * d) return;
  const int p = (int)(e % B), k = (int)(e / B);
  Xb[e] = X[(size_t)k * ldx + batch[p]];
}

// bsum[b] = dist[b*1024] + dist[b*1024 + 1] + ... left to right (one thread per block of 1024)
__global__ void mb_sse_blocks_kernel(const double *__restrict__ dist, int B, double *__restrict__ bsum, const int *__restrict__ state) {
  if (state[2]) return;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int nblk = (B + MB_BLK - 1) / MB_BLK;
  if (b >= nblk) return;
  const int i0 = b * MB_BLK, i1 = min(i0 + MB_BLK, B);
  double acc = 0.0;
  for (int q = i0; q < i1; ++q) acc = acc + dist[q];
  bsum[b] = acc;
}

// sse = the block sums left to right; early-stop bookkeeping: state = {iterations run, stall, stop}, fstate = {best sse}
__global__ void mb_sse_final_kernel(const double *__restrict__ bsum, int nblk, int early_stop, int *__restrict__ state,
                                    double *__restrict__ fstate) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || state[2]) return;
  double sse = 0.0;
  for (int b = 0; b < nblk; ++b) sse = sse + bsum[b];
  state[0] += 1;
  if (sse < fstate[0]) { fstate[0] = sse; state[1] = 0; } else { state[1] += 1; }
  if (state[1] >= early_stop) state[3] = 1;         // takes effect AFTER this iteration's update (latched below)
}

// per centre c, coordinate k: its batch points in batch order:  v += 1; eta = 1 / v; c_k = (1 - eta) c_k + eta x_k
__global__ __launch_bounds__(64) void mb_update_kernel(const double *__restrict__ Xb, int B, const int *__restrict__ colptr,
                                                       const int *__restrict__ pos, int s, double *__restrict__ C,
                                                       const double *__restrict__ cnt_in, double *__restrict__ cnt_out,
                                                       const int *__restrict__ state) {
  if (state[2]) return;
  const int c = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (c >= s) return;
  const int p0 = colptr[c], p1 = colptr[c + 1];
  double v = cnt_in[c], ck = C[(size_t)k * s + c];
  const double *xk = Xb + (size_t)k * B;
  for (int p = p0; p < p1; ++p) {
    v = v + 1.0;
    const double eta = 1.0 / v;
    const double a = (1.0 - eta) * ck, b2 = eta * xk[pos[p]];
    ck = a + b2;
  }
  C[(size_t)k * s + c] = ck;
  if (k == 0) cnt_out[c] = v;
}

// after the update: a pending stop becomes effective
__global__ void mb_latch_kernel(int *__restrict__ state) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && state[3]) state[2] = 1;
}

// SplitMix64 counter stream: the q-th draw of stream `st` under `seed`
static inline unsigned long long mb_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline double mb_uniform(unsigned long long seed, unsigned long long st, unsigned long long q) {
  const unsigned long long h = mb_mix(mb_mix(seed + 0x632BE59BD9B4E019ull * (st + 1)) + q);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);      // [0, 1)
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_dev_anchor_dpad(int d);
extern "C" int flgp_dev_anchor_rows(int s);
extern "C" int flgp_dev_anchor_prep(void *stream, const double *dU, int s, int ldu, int d, double *dUt, double *duu);
extern "C" int flgp_dev_knn(void *stream, const double *dX, int n, int ldx, int d, const double *dUt, const double *duu,
                            int s, int r, int *d_idx, double *d_dist, int ldo);
extern "C" int flgp_dev_mean(void *stream, const double *d_x, long count, double *d_out, double *d_work);

namespace {
__global__ void mb_count_kernel(const int *__restrict__ lab, long n, int s, int *__restrict__ cnt) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int q = lab[i];
    if (q >= 0 && q < s) atomicAdd(&cnt[q], 1);
  }
}
__global__ void mb_sizes_kernel(const int *__restrict__ cnt, int s, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < s) out[i] = (double)cnt[i];
}
}  // namespace

extern "C" size_t flgp_dev_csc_workspace(int n, int s, int r);
extern "C" int flgp_dev_csc_build(void *stream, const int *d_ell_idx, int n, int s, int r, int *d_colptr, int *d_pos,
                                  void *d_work, size_t work_bytes);

extern "C" int flgp_kmeans_minibatch(const double *X, int n, int d, int s, int batch_size, int num_init, int max_iters,
                                     double init_fraction, int early_stop_iter, unsigned long long seed, double *U_out,
                                     int *info, double *withinss_out) {
  FLGP_REQUIRE(X && U_out, "kmeans_minibatch: null pointer");
  FLGP_REQUIRE(n >= 1 && d >= 1 && s >= 1 && s <= n, "kmeans_minibatch: need 1 <= s <= n");
  // the reference's arguments (src/Utils.cpp:52-55) as defaults: batch_size = 10 s, init_fraction = 20 s / n
  if (batch_size <= 0) batch_size = (int)std::min<long>((long)n, 10L * s);
  if (init_fraction <= 0.0) init_fraction = std::min(1.0, 20.0 * (double)s / (double)n);
  FLGP_REQUIRE(batch_size >= 1 && batch_size <= n && num_init >= 1 && max_iters >= 1 && early_stop_iter >= 1,
               "kmeans_minibatch: need 1 <= batch_size <= n, num_init >= 1, max_iters >= 1, early_stop_iter >= 1");
  FLGP_REQUIRE(init_fraction > 0.0 && init_fraction <= 1.0, "kmeans_minibatch: init_fraction must lie in (0, 1]");
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0, "kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  hipStream_t st = nullptr;
  FLGP_HIP(hipStreamCreate(&st));
  struct Closer { hipStream_t s; ~Closer() { (void)hipStreamDestroy(s); } } closer{st};
  long nsub_l = (long)std::ceil(init_fraction * (double)n);
  if (nsub_l < s) nsub_l = s;
  if (nsub_l > n) nsub_l = n;
  const int B = batch_size, nbb = ceil_div(B, MB_BLK);
  const int nsub = (int)nsub_l, nblk = ceil_div(nsub, MB_BLK), rows = flgp_dev_anchor_rows(s);
  DevBuf dX, dC, dbest, dsub, d2, bsum, cnew, picked, dbatch, cnt[2], dstate, fstate, Ut, uu, lab, dist, mwork, mean, icnt, Xb, blab, bdist,
      colptr, pos, cwork;
  FLGP_TRY(dX.alloc(sizeof(double) * (size_t)n * d));
  FLGP_TRY(dC.alloc(sizeof(double) * (size_t)s * (d + 1)));
  FLGP_TRY(dbest.alloc(sizeof(double) * (size_t)s * (d + 1)));
  FLGP_TRY(dsub.alloc(sizeof(int) * (size_t)nsub));
  FLGP_TRY(d2.alloc(sizeof(double) * (size_t)nsub));
  FLGP_TRY(bsum.alloc(sizeof(double) * (size_t)std::max(nblk, nbb)));
  FLGP_TRY(cnew.alloc(sizeof(double) * (size_t)d));
  FLGP_TRY(picked.alloc(sizeof(int) * (size_t)s));
  FLGP_TRY(dbatch.alloc(sizeof(int) * (size_t)B));
  FLGP_TRY(cnt[0].alloc(sizeof(double) * (size_t)s)); FLGP_TRY(cnt[1].alloc(sizeof(double) * (size_t)s));
  FLGP_TRY(dstate.alloc(sizeof(int) * 4)); FLGP_TRY(fstate.alloc(sizeof(double)));
  FLGP_TRY(Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(lab.alloc(sizeof(int) * (size_t)n));
  FLGP_TRY(dist.alloc(sizeof(double) * (size_t)n));
  FLGP_TRY(mwork.alloc(sizeof(double) * (size_t)(ceil_div((long)n, 4096) + 1)));
  FLGP_TRY(mean.alloc(sizeof(double)));
  FLGP_TRY(icnt.alloc(sizeof(int) * (size_t)s));
  FLGP_TRY(Xb.alloc(sizeof(double) * (size_t)B * d));
  FLGP_TRY(blab.alloc(sizeof(int) * (size_t)B)); FLGP_TRY(bdist.alloc(sizeof(double) * (size_t)B));
  FLGP_TRY(colptr.alloc(sizeof(int) * (size_t)(s + 1))); FLGP_TRY(pos.alloc(sizeof(int) * (size_t)B));
  const size_t cwb = flgp_dev_csc_workspace(B, s, 1);
  FLGP_TRY(cwork.alloc(cwb));
  FLGP_HIP(hipMemcpyAsync(dX.p, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, st));
  std::vector<int> perm((size_t)n);
  double best = 0.0;
  int best_it = 0, best_start = 0;
  for (int q = 0; q < num_init; ++q) {
    const unsigned long long base = (unsigned long long)q * 4;
    // ---- subsample: partial Fisher-Yates, draw i swaps position i with position i + floor(u (n - i))
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int i = 0; i < nsub; ++i) {
      long j = i + (long)(mb_uniform(seed, base + 0, (unsigned long long)i) * (double)(n - i));
      if (j >= n) j = n - 1;
      std::swap(perm[i], perm[j]);
    }
    FLGP_HIP(hipMemcpyAsync(dsub.p, perm.data(), sizeof(int) * (size_t)nsub, hipMemcpyHostToDevice, st));
    FLGP_HIP(hipStreamSynchronize(st));      // (perm is shuffled further below)
    // ---- k-means++ on the subsample
    int first = (int)(mb_uniform(seed, base + 1, 0) * (double)nsub);
    if (first >= nsub) first = nsub - 1;
    double *C = dC.as<double>();
    hipLaunchKernelGGL(mb_pp_first_kernel, dim3(1), dim3(64), 0, st, dX.as<double>(), n, d, dsub.as<int>(), first, C, s, cnew.as<double>(),
                       picked.as<int>());
    for (int c = 1; c < s; ++c) {
      hipLaunchKernelGGL(mb_pp_update_kernel, dim3(nblk), dim3(MB_BLK), 0, st, dX.as<double>(), n, d, dsub.as<int>(), nsub,
                         cnew.as<double>(), c == 1 ? 1 : 0, d2.as<double>(), bsum.as<double>());
      hipLaunchKernelGGL(mb_pp_pick_kernel, dim3(1), dim3(64), 0, st, dX.as<double>(), n, d, dsub.as<int>(), nsub, d2.as<double>(),
                         bsum.as<double>(), nblk, mb_uniform(seed, base + 1, (unsigned long long)c), C, s, c, cnew.as<double>(),
                         picked.as<int>());
    }
    FLGP_TRY(check_launch("mb_pp kernels"));
    // ---- mini-batch iterations.  Batch of iteration `it`: the first B positions of the running permutation after B more
    //      Fisher-Yates draws (draw p swaps position p with p + floor(u (n - p))): B distinct rows.  The early stop is
    //      decided on the device (state[2]); the launches after it do nothing.
    FLGP_HIP(hipMemsetAsync(cnt[0].p, 0, sizeof(double) * (size_t)s, st));
    FLGP_HIP(hipMemsetAsync(dstate.p, 0, sizeof(int) * 4, st));
    const double inf = INFINITY;
    FLGP_HIP(hipMemcpyAsync(fstate.p, &inf, sizeof(double), hipMemcpyHostToDevice, st));
    FLGP_HIP(hipStreamSynchronize(st));
    unsigned long long draw = 0;
    for (int it = 0; it < max_iters; ++it) {
      for (int p = 0; p < B; ++p) {
        long j = p + (long)(mb_uniform(seed, base + 2, draw++) * (double)(n - p));
        if (j >= n) j = n - 1;
        std::swap(perm[p], perm[j]);
      }
      FLGP_HIP(hipMemcpyAsync(dbatch.p, perm.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice, st));
      // the stop flag of the iterations so far comes back with the same synchronisation (looked at only every
      // eighth iteration, up to seven idle iterations -- k-NN pass, CSC build, upload -- were enqueued behind the latch)
      int hs[4] = {0, 0, 0, 0};
      if (it > 0) FLGP_HIP(hipMemcpyAsync(hs, dstate.p, sizeof(hs), hipMemcpyDeviceToHost, st));
      FLGP_HIP(hipStreamSynchronize(st));    // (the next iteration shuffles perm again)
      if (hs[2]) break;
      const int *state = dstate.as<int>();
      hipLaunchKernelGGL(mb_gather_kernel, dim3(ceil_div((long)B * d, 256)), dim3(256), 0, st, dX.as<double>(), n, d, dbatch.as<int>(), B,
                         Xb.as<double>(), state);
      FLGP_TRY(flgp_dev_anchor_prep(st, C, s, s, d, Ut.as<double>(), uu.as<double>()));
      FLGP_TRY(flgp_dev_knn(st, Xb.as<double>(), B, B, d, Ut.as<double>(), uu.as<double>(), s, 1, blab.as<int>(), bdist.as<double>(), B));
      hipLaunchKernelGGL(mb_sse_blocks_kernel, dim3(ceil_div(nbb, 64)), dim3(64), 0, st, bdist.as<double>(), B, bsum.as<double>(), state);
      hipLaunchKernelGGL(mb_sse_final_kernel, dim3(1), dim3(64), 0, st, bsum.as<double>(), nbb, early_stop_iter, dstate.as<int>(),
                         fstate.as<double>());
      FLGP_TRY(flgp_dev_csc_build(st, blab.as<int>(), B, s, 1, colptr.as<int>(), pos.as<int>(), cwork.p, cwb));
      hipLaunchKernelGGL(mb_update_kernel, dim3(ceil_div(s, 64), d), dim3(64), 0, st, Xb.as<double>(), B, colptr.as<int>(), pos.as<int>(), s, C,
                         cnt[it & 1].as<double>(), cnt[(it + 1) & 1].as<double>(), state);
      hipLaunchKernelGGL(mb_latch_kernel, dim3(1), dim3(64), 0, st, dstate.as<int>());
      FLGP_TRY(check_launch("mini-batch iteration"));
    }
    // ---- evaluation over all points: 1-NN labels, within-SS, sizes
    FLGP_TRY(flgp_dev_anchor_prep(st, C, s, s, d, Ut.as<double>(), uu.as<double>()));
    FLGP_TRY(flgp_dev_knn(st, dX.as<double>(), n, n, d, Ut.as<double>(), uu.as<double>(), s, 1, lab.as<int>(), dist.as<double>(), n));
    FLGP_TRY(flgp_dev_mean(st, dist.as<double>(), n, mean.as<double>(), mwork.as<double>()));
    FLGP_HIP(hipMemsetAsync(icnt.p, 0, sizeof(int) * (size_t)s, st));
    hipLaunchKernelGGL(mb_count_kernel, dim3((unsigned)std::min<long>(2048, ((long)n + 255) / 256)), dim3(256), 0, st, lab.as<int>(), (long)n, s,
                       icnt.as<int>());
    hipLaunchKernelGGL(mb_sizes_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, icnt.as<int>(), s, C + (size_t)s * d);
    FLGP_TRY(check_launch("mb_count_kernel"));
    double m = 0.0; int hinfo[4] = {0, 0, 0, 0};
    FLGP_HIP(hipMemcpyAsync(&m, mean.p, sizeof(double), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipMemcpyAsync(hinfo, dstate.p, sizeof(hinfo), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    const double wss = m * (double)n;
    if (q == 0 || wss < best) {
      best = wss; best_it = hinfo[0]; best_start = q;
      FLGP_HIP(hipMemcpyAsync(dbest.p, dC.p, sizeof(double) * (size_t)s * (d + 1), hipMemcpyDeviceToDevice, st));
    }
  }
  FLGP_HIP(hipMemcpyAsync(U_out, dbest.p, sizeof(double) * (size_t)s * (d + 1), hipMemcpyDeviceToHost, st));
  FLGP_HIP(hipStreamSynchronize(st));
  if (info) { info[0] = best_it; info[1] = best_start; }
  if (withinss_out) *withinss_out = best;
  return FLGP_OK;
}
