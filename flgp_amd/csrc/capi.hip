// Host-pointer entry points of libflgp_hip.so: the .Call boundary (include/flgp_hip.h).
// Each one stages R-owned host buffers into HBM, runs the device stages on one stream and
// copies the result back; no state survives the call (as the reference: no handles/caches).
#include "common.h"
#include <chrono>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

using namespace flgp;

extern "C" int flgp_dev_anchor_rows(int s);
extern "C" int flgp_dev_v_to_z(void *stream, const double *d_v, int r, double *d_z);
extern "C" size_t flgp_dev_hk_workspace(int n0, int n1, int K, int gather0);
extern "C" int flgp_dev_mean(void *stream, const double *d_x, long count, double *d_out, double *d_work);
extern "C" int flgp_dev_col_scale_row_normalize(void *stream, const int *d_ell_idx, double *d_ell_val, int n, int r,
                                                const double *d_colsum, const double *d_num_class);
extern "C" int flgp_dev_se_weights_den(void *stream, const int *d_knn_idx, const double *d_knn_dist, int n, int ldk,
                                       int r, double den, int *d_ell_idx, double *d_ell_val);


// u = A v / sigma (flgp_dev_u_recover) needs sigma > 0 for every wanted pair.  K == s on a rank-deficient A, an anchor no
// point chose, an SE bandwidth that underflows a column: the Gram route cannot deliver those left vectors (the
// reference's BDCSVD can), and the recovery kernel writes a zero column for sigma == 0 rather than Inf / NaN.  EVERY
// caller of flgp_dev_eig_topk that goes on to flgp_dev_u_recover asks here first (the host spectrum did, the
// bandwidth grid and the sharded driver did not): FLGP_OK, or FLGP_ERR_INVALID / FLGP_ERR_NOCONV with the message.
// Synchronises the stream (flgp_dev_eig_topk has done so already: the copy is K doubles).
extern "C" int flgp_dev_spectrum_usable_route(void *stream, const double *d_eig, int K, int full_decomposition) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(d_eig && K >= 1, "spectrum_usable: bad arguments");
  std::vector<double> hv((size_t)K);
  FLGP_HIP(hipMemcpyAsync(hv.data(), d_eig, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, st));
  FLGP_HIP(hipStreamSynchronize(st));
  const double top = hv[0], low = hv[K - 1];
  if (!(top > 0.0) || !std::isfinite(top)) { set_error("spectrum: the similarity matrix is zero or not finite"); return FLGP_ERR_INVALID; }
  // What the solver delivered is what bounds the left vectors u = A v / sigma: the block solver stops at
  // residuals of 5e-11 lambda_1, so an eigenvalue below ~1e-10 lambda_1 is rounding noise and its u noise amplified by
  // lambda_1 / lambda; the full decomposition (K == s: Jacobi to working precision) resolves eigenvalues down to
  // ~1e3 eps lambda_1.  Below that the Gram route has nothing to offer -- the reference's SVD of A itself would.
  const double floor_rel = full_decomposition ? std::pow(10.0, -(double)tuning("spectrum_floor_exp_full", 13))
                                              : std::pow(10.0, -(double)tuning("spectrum_floor_exp", 10));
  if (!(low > floor_rel * top)) {
    set_error("spectrum: sigma_%d^2 = %.3e is below %.0e sigma_1^2 (= %.3e), what the %s resolves: K = %d reaches into the "
              "(numerical) null space of the similarity matrix, whose left vectors the Gram route cannot recover -- choose a "
              "smaller K", K, low, floor_rel, top, full_decomposition ? "full decomposition" : "block eigensolver", K);
    return FLGP_ERR_NOCONV;
  }
  return FLGP_OK;
}
extern "C" int flgp_dev_spectrum_usable(void *stream, const double *d_eig, int K) {
  return flgp_dev_spectrum_usable_route(stream, d_eig, K, 0);
}

namespace {

struct Stream {
  hipStream_t s = nullptr;
  ~Stream() { if (s) (void)hipStreamDestroy(s); }
  int create() { FLGP_HIP(hipStreamCreate(&s)); return FLGP_OK; }
};

int check_distance(const char *distance) {
  if (distance && !strcmp(distance, "Euclidean")) return FLGP_OK;
  // reference: Rcpp::stop("The distance method of KNN is not supported!\n"), src/Utils.cpp:123
  set_error("The distance method of KNN is not supported!");
  return FLGP_ERR_UNSUPPORTED;
}

// Input validation for the host entry points, on the device (the data is there anyway; a pass over 128 MB is 30 us).
// flag |= 1 for a non-finite coordinate, |= 2 for a column index outside [0, s).
__global__ void check_finite_kernel(const double *__restrict__ x, long count, int *__restrict__ flag) {
  const long stride = (long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    const double v = x[e];
    bad |= !(__builtin_fabs(v) <= 1.7976931348623157e308);     // false for NaN and +-Inf
  }
  if (bad) atomicOr(flag, 1);
}
__global__ void check_index_kernel(const int *__restrict__ idx, long count, int s, int *__restrict__ flag) {
  const long stride = (long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) bad |= (unsigned)idx[e] >= (unsigned)s;
  if (bad) atomicOr(flag, 2);
}
struct InputCheck {
  DevBuf flag;
  int begin(hipStream_t st) {
    FLGP_TRY(flag.alloc(sizeof(int)));
    FLGP_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    return FLGP_OK;
  }
  int finite(hipStream_t st, const double *d_x, long count) {
    if (count <= 0) return FLGP_OK;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)std::min<long>(2048, (count + 255) / 256)), dim3(256), 0, st, d_x, count, flag.as<int>());
    return check_launch("check_finite_kernel");
  }
  int indices(hipStream_t st, const int *d_idx, long count, int s) {
    if (count <= 0) return FLGP_OK;
    hipLaunchKernelGGL(check_index_kernel, dim3((unsigned)std::min<long>(2048, (count + 255) / 256)), dim3(256), 0, st, d_idx, count, s, flag.as<int>());
    return check_launch("check_index_kernel");
  }
  // synchronises; FLGP_ERR_INVALID with the reason
  int verdict(hipStream_t st, const char *who) {
    int h = 0;
    FLGP_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    if (h & 1) { set_error("%s: the input holds NA / NaN / Inf values (the reference returns garbage here; this library refuses)", who); return FLGP_ERR_INVALID; }
    if (h & 2) { set_error("%s: a column index of the sparse matrix is outside [0, s)", who); return FLGP_ERR_INVALID; }
    return FLGP_OK;
  }
};

int h2d(void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes) FLGP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
  return FLGP_OK;
}
int d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes) FLGP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
  return FLGP_OK;
}

// device-side state of one similarity matrix: anchors, k-NN, ELL (+ CSC view)
struct Sim {
  DevBuf X, U, Ut, uu, knn_idx, knn_dist, ell_idx, ell_val, colptr, pos, colsum, cswork, work, num_class;
  int n = 0, d = 0, s = 0, r = 0;
  bool have_csc = false;
  // row-sharded runs (flgp_dev_heat_kernel_covariance_sharded): n is the LOCAL row count, ldx the leading dimension of
  // the caller's X block, comm carries the exchanges of SURVEY 8e, n_global scales the eigenvectors, sizes overrides U's
  // last column as the cluster sizes
  int ldx = 0;
  const flgp_comm *comm = nullptr;
  long n_global = 0;
  const double *sizes = nullptr;
  // the CSC view depends on the PATTERN alone, the Laplacian passes only touch the values: the view is built on a second
  // stream beside them (round 4) and joined before the Gram kernel, its first reader
  hipEvent_t csc_ev = nullptr;
  bool csc_pending = false, want_csc = false;
  ~Sim() { if (csc_ev) { (void)hipEventSynchronize(csc_ev); (void)hipEventDestroy(csc_ev); } }
};

int upload_points(Sim &S, hipStream_t st, const double *X, int n, int d, const double *U, int s, int ucols,
                  bool need_sizes) {
  FLGP_REQUIRE(X && U, "null pointer");
  FLGP_REQUIRE(n >= 1 && d >= 1 && s >= 1, "bad shape n=%d d=%d s=%d", n, d, s);
  FLGP_REQUIRE(ucols == d || ucols == d + 1, "U must have d or d+1 columns (d=%d, got %d)", d, ucols);
  FLGP_REQUIRE(!need_sizes || ucols == d + 1,
               "gl=\"cluster-normalized\" needs the cluster sizes in column d+1 of U (the reference reads out of bounds here)");
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0, "kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  S.n = n; S.d = d; S.s = s;
  const int rows = flgp_dev_anchor_rows(s);
  FLGP_TRY(S.X.alloc(sizeof(double) * (size_t)n * d));
  FLGP_TRY(S.U.alloc(sizeof(double) * (size_t)s * ucols));
  FLGP_TRY(S.Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(S.uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(h2d(S.X.p, X, sizeof(double) * (size_t)n * d, st));
  FLGP_TRY(h2d(S.U.p, U, sizeof(double) * (size_t)s * ucols, st));
  {   // a NaN row would leave the k-NN lists empty (knn.hip); refuse before anything is computed from it
    InputCheck ck;
    FLGP_TRY(ck.begin(st));
    FLGP_TRY(ck.finite(st, S.X.as<double>(), (long)n * d));
    FLGP_TRY(ck.finite(st, S.U.as<double>(), (long)s * ucols));
    FLGP_TRY(ck.verdict(st, "points / anchors"));
  }
  return flgp_dev_anchor_prep(st, S.U.as<double>(), s, s, d, S.Ut.as<double>(), S.uu.as<double>());
}

int run_knn(Sim &S, hipStream_t st, int r, bool want_dist) {
  FLGP_REQUIRE(r >= 1 && r <= S.s, "need 1 <= r <= s (r=%d, s=%d)", r, S.s);
  S.r = r;
  FLGP_TRY(S.knn_idx.alloc(sizeof(int) * (size_t)S.n * r));
  if (want_dist) FLGP_TRY(S.knn_dist.alloc(sizeof(double) * (size_t)S.n * r));
  return flgp_dev_knn(st, S.X.as<double>(), S.n, S.ldx ? S.ldx : S.n, S.d, S.Ut.as<double>(), S.uu.as<double>(), S.s, r,
                      S.knn_idx.as<int>(), want_dist ? S.knn_dist.as<double>() : nullptr, S.n);
}

int alloc_ell(Sim &S) {
  FLGP_TRY(S.ell_idx.alloc(sizeof(int) * (size_t)S.n * S.r));
  return S.ell_val.alloc(sizeof(double) * (size_t)S.n * S.r);
}

// column sums of the device ELL into S.colsum (two-level order of the oracle)
int colsum_of(Sim &S, hipStream_t st, const int *d_idx, const double *d_val) {
  const size_t wb = flgp_dev_colsum_workspace(S.n, S.s);
  if (!S.colsum.p) FLGP_TRY(S.colsum.alloc(sizeof(double) * (size_t)S.s));
  if (!S.cswork.p) FLGP_TRY(S.cswork.alloc(wb));
  return flgp_dev_colsum(st, d_idx, d_val, S.n, S.r, S.s, S.colsum.as<double>(), S.cswork.p, wb);
}

int build_csc(Sim &S, hipStream_t st) {
  if (S.have_csc) return FLGP_OK;
  const size_t wb = flgp_dev_csc_workspace(S.n, S.s, S.r);
  FLGP_TRY(S.work.alloc(wb));
  FLGP_TRY(S.colptr.alloc(sizeof(int) * (size_t)(S.s + 1)));
  FLGP_TRY(S.pos.alloc(sizeof(int) * (size_t)S.n * S.r));
  FLGP_TRY(flgp_dev_csc_build(st, S.ell_idx.as<int>(), S.n, S.s, S.r, S.colptr.as<int>(), S.pos.as<int>(), S.work.p, wb));
  S.have_csc = true;
  return FLGP_OK;
}

// one second stream per host thread and device for work that may run beside the caller's stream (never destroyed)
static hipStream_t side_stream() {
  thread_local hipStream_t side[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!side[dev] && hipStreamCreateWithFlags(&side[dev], hipStreamNonBlocking) != hipSuccess) side[dev] = nullptr;
  return side[dev];
}

// the CSC view on the second stream, ordered behind what `st` holds now; join_csc() makes `st` wait for it
int build_csc_async(Sim &S, hipStream_t st) {
  if (S.have_csc) return FLGP_OK;
  hipStream_t sd = tuning("csc_async", 1) ? side_stream() : nullptr;
  if (!sd) return build_csc(S, st);
  if (!S.csc_ev) FLGP_HIP(hipEventCreateWithFlags(&S.csc_ev, hipEventDisableTiming));
  FLGP_HIP(hipEventRecord(S.csc_ev, st));
  FLGP_HIP(hipStreamWaitEvent(sd, S.csc_ev, 0));
  FLGP_TRY(build_csc(S, sd));
  FLGP_HIP(hipEventRecord(S.csc_ev, sd));
  S.csc_pending = true;
  return FLGP_OK;
}
int join_csc(Sim &S, hipStream_t st) {
  if (S.csc_pending) { FLGP_HIP(hipStreamWaitEvent(st, S.csc_ev, 0)); S.csc_pending = false; }
  return FLGP_OK;
}

// graphLaplacian_cpp on the device ELL (reference src/Utils.cpp:195-212)
int laplacian(Sim &S, hipStream_t st, int gl, const double *d_num_class) {
  if (gl != FLGP_GL_RW) {
    FLGP_TRY(colsum_of(S, st, S.ell_idx.as<int>(), S.ell_val.as<double>()));
    FLGP_TRY(flgp_comm_all_reduce_sum(S.comm, S.colsum.as<double>(), (size_t)S.s, st));            // exchange 2a
    return flgp_dev_col_scale_row_normalize(st, S.ell_idx.as<int>(), S.ell_val.as<double>(), S.n, S.r, S.colsum.as<double>(),
                                            gl == FLGP_GL_CLUSTER_NORMALIZED ? d_num_class : nullptr);     // one pass over the values
  }
  return flgp_dev_row_normalize(st, S.ell_val.as<double>(), S.n, S.r);
}

// cross_similarity_{lae,se}_cpp after the points are on the device (reference src/Spectrum.cpp:101-142)
int cross_similarity(Sim &S, hipStream_t st, int r, int kernel_se, int gl, double epsilon, int ucols) {
  FLGP_TRY(run_knn(S, st, r, kernel_se != 0));
  FLGP_TRY(alloc_ell(S));
  if (kernel_se)
    FLGP_TRY(flgp_dev_se_weights(st, S.knn_idx.as<int>(), S.knn_dist.as<double>(), S.n, S.n, r, epsilon,
                                 S.ell_idx.as<int>(), S.ell_val.as<double>()));
  else
    FLGP_TRY(flgp_dev_lae(st, S.X.as<double>(), S.n, S.ldx ? S.ldx : S.n, S.d, S.Ut.as<double>(), S.s, r, S.knn_idx.as<int>(), S.n,
                          S.ell_idx.as<int>(), S.ell_val.as<double>()));
  if (gl < 0) return FLGP_OK;  // LAE_cpp alone: no graph-Laplacian normalisation
  if (S.want_csc) FLGP_TRY(build_csc_async(S, st));     // (callers that go on to the spectrum: beside the Laplacian passes)
  const double *sizes = S.sizes ? S.sizes : ((ucols == S.d + 1) ? S.U.as<double>() + (size_t)S.d * S.s : nullptr);  // U.col(d)
  return laplacian(S, st, gl, sizes);
}

int csr_out(Sim &S, hipStream_t st, int *csr_p, int *csr_j, double *csr_x) {
  FLGP_TRY(d2h(csr_j, S.ell_idx.p, sizeof(int) * (size_t)S.n * S.r, st));
  FLGP_TRY(d2h(csr_x, S.ell_val.p, sizeof(double) * (size_t)S.n * S.r, st));
  FLGP_HIP(hipStreamSynchronize(st));
  if (csr_p)
    for (long i = 0; i <= S.n; ++i) csr_p[i] = (int)(i * S.r);
  return FLGP_OK;
}

// spectrum_from_Z_cpp on the device ELL (reference src/Spectrum.cpp:146-161): leaves values (K)
// and vectors (n x K) on the device
struct Spectrum {
  DevBuf G, eig, V, values, vectors, work, uwork;
  int K = 0;
};

int spectrum(Sim &S, hipStream_t st, int K, int root, Spectrum &P, int *info) {
  if (K < 0) K = S.s;   // reference: K < 0 -> s (src/Spectrum.cpp:31-33,69-71; src/TruncatedSVD.cpp:11-13)
  FLGP_REQUIRE(K >= 1 && K <= S.s, "need 1 <= K <= s (K=%d, s=%d)", K, S.s);
  P.K = K;
  FLGP_TRY(build_csc_async(S, st));     // (no-op when cross_similarity has started it already)
  // A = Z diag(1/sqrt(|colsum|+1e-9))  (:149-150)
  FLGP_TRY(colsum_of(S, st, S.ell_idx.as<int>(), S.ell_val.as<double>()));
  FLGP_TRY(flgp_comm_all_reduce_sum(S.comm, S.colsum.as<double>(), (size_t)S.s, st));              // exchange 2b
  FLGP_TRY(flgp_dev_col_scale(st, S.ell_idx.as<int>(), S.ell_val.as<double>(), S.n, S.r, S.colsum.as<double>(), nullptr, 1));
  // Gram + top-K eigenpairs (replaces RSpectra::svds / BDCSVD, src/TruncatedSVD.cpp:17-30)
  FLGP_TRY(P.G.alloc(sizeof(double) * (size_t)S.s * S.s));
  FLGP_TRY(join_csc(S, st));
  FLGP_TRY(flgp_dev_gram(st, S.ell_idx.as<int>(), S.ell_val.as<double>(), S.n, S.s, S.r, S.colptr.as<int>(),
                         S.pos.as<int>(), P.G.as<double>(), S.s));
  if (flgp_comm_world(S.comm) > 1) {
    // exchange 3: the Gram partials of the row blocks.  Only the upper triangle travels (s(s+1)/2 doubles); unpacking
    // mirrors it, which also leaves G symmetric bit for bit whatever order the transport added the ranks in.  The
    // eigensolve below then runs replicated: every rank holds the same G.
    DevBuf packed;
    const size_t cnt = (size_t)S.s * ((size_t)S.s + 1) / 2;
    FLGP_TRY(packed.alloc(sizeof(double) * cnt));
    FLGP_TRY(flgp_dev_sym_pack(st, P.G.as<double>(), S.s, S.s, packed.as<double>()));
    FLGP_TRY(flgp_comm_all_reduce_sum(S.comm, packed.as<double>(), cnt, st));
    FLGP_TRY(flgp_dev_sym_unpack(st, packed.as<double>(), S.s, P.G.as<double>(), S.s));
    FLGP_HIP(hipStreamSynchronize(st));     // (packed dies here)
  }
  const size_t wb = flgp_dev_eig_workspace(S.s, K);
  FLGP_TRY(P.work.alloc(wb));
  FLGP_TRY(P.eig.alloc(sizeof(double) * (size_t)K));
  FLGP_TRY(P.V.alloc(sizeof(double) * (size_t)S.s * K));
  int solve_info[4] = {0, 0, 0, 0};
  FLGP_TRY(flgp_dev_eig_topk(st, P.G.as<double>(), S.s, S.s, K, 0.0, P.eig.as<double>(), P.V.as<double>(), S.s,
                             P.work.p, wb, solve_info));
  if (info) for (int q = 0; q < 4; ++q) info[q] = solve_info[q];
  FLGP_TRY(flgp_dev_spectrum_usable_route(st, P.eig.as<double>(), K, solve_info[2]));
  // u = A v / sigma, vectors = u sqrt(n), values = sigma^2 (or sigma if root)  (:153-158)
  FLGP_TRY(P.values.alloc(sizeof(double) * (size_t)K));
  FLGP_TRY(P.vectors.alloc(sizeof(double) * (size_t)S.n * K));
  FLGP_TRY(P.uwork.alloc(flgp_dev_u_recover_workspace(S.s, K)));
  return flgp_dev_u_recover(st, S.ell_idx.as<int>(), S.ell_val.as<double>(), S.n, S.r, P.V.as<double>(), S.s, S.s,
                            P.eig.as<double>(), K, std::sqrt((double)(S.n_global ? S.n_global : (long)S.n)), root,
                            P.vectors.as<double>(), S.n, P.values.as<double>(), P.uwork.as<double>());
}

int parse_kernel(const char *kernel, int *se) {
  if (kernel && !strcmp(kernel, "lae")) { *se = 0; return FLGP_OK; }
  if (kernel && !strcmp(kernel, "se")) { *se = 1; return FLGP_OK; }
  // the reference only prints and carries on with an empty Z (src/Spectrum.cpp:65-67); here it is an error
  set_error("The kernel type is not supported!");
  return FLGP_ERR_UNSUPPORTED;
}

bool is_range(const int *idx, int cnt) {
  for (int i = 1; i < cnt; ++i)
    if (idx[i] != idx[0] + i) return false;
  return true;
}

// ------------------------------------------------------------------------------------------
// H to the caller's (pageable) buffer without serialising GEMM, PCIe and the host copy.
// H is n0 x n1 column-major and the contraction is independent per column, so H goes over in blocks of columns:
// block c is contracted into one of two device buffers while block c-1 crosses PCIe into one of two pinned buffers and
// block c-2 is copied from there into the caller's memory by a few host threads.  (hipMemcpyAsync straight into
// pageable memory stages through the runtime's own bounce buffer on ONE thread and never overlaps the GEMM: 0.44-0.74 s
// for the 8 GB of BASELINE configs[2] in round 1, against 0.16 s of PCIe.)  The pinned buffers are kept for the
// lifetime of the process (pinning 1 GB costs more than the whole call).
// ------------------------------------------------------------------------------------------
struct PinnedRing {
  void *buf[2] = {nullptr, nullptr};
  size_t bytes = 0;
  bool busy = false;
  void drop() {
    for (int q = 0; q < 2; ++q) { if (buf[q]) (void)hipHostFree(buf[q]); buf[q] = nullptr; }
    bytes = 0;
  }
  int ensure(size_t need) {
    if (need <= bytes) return FLGP_OK;
    drop();
    for (int q = 0; q < 2; ++q)
      if (hipHostMalloc(&buf[q], need, hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc of %zu bytes failed", need);
        drop();
        return FLGP_ERR_NOMEM;
      }
    bytes = need;
    return FLGP_OK;
  }
};
// The rings are handed out one per call in flight (the lock covers the hand-out only, not the multi-GB transfer): a
// second caller gets a ring of its own, up to `hk_rings_max` (2); beyond that callers queue.  flgp_release_pinned()
// gives the idle ones back to the system.
static std::mutex g_ring_mu;
static std::condition_variable g_ring_cv;
static std::vector<PinnedRing *> g_rings;
struct RingLease {
  PinnedRing *r = nullptr;
  explicit RingLease(int at_least = 0) {     // at_least: the ranks of one multi-GPU call each need a ring at the same time
    std::unique_lock<std::mutex> lk(g_ring_mu);
    const size_t cap = (size_t)std::max(std::max(1, tuning("hk_rings_max", 2)), at_least);
    for (;;) {
      for (PinnedRing *c : g_rings) if (!c->busy) { r = c; break; }
      if (!r && g_rings.size() < cap) { r = new PinnedRing(); g_rings.push_back(r); }
      if (r) break;
      g_ring_cv.wait(lk);
    }
    r->busy = true;
  }
  ~RingLease() {
    { std::lock_guard<std::mutex> lk(g_ring_mu); r->busy = false; }
    g_ring_cv.notify_one();
  }
};
// events of one call, destroyed on every way out
struct EventSet {
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
  int create() {
    for (int q = 0; q < 4; ++q)
      if (hipEventCreateWithFlags(&e[q], hipEventDisableTiming) != hipSuccess) { e[q] = nullptr; set_error("hipEventCreate failed"); return FLGP_ERR_HIP; }
    return FLGP_OK;
  }
  ~EventSet() { for (int q = 0; q < 4; ++q) if (e[q]) (void)hipEventDestroy(e[q]); }
};

static void parallel_copy(char *dst, const char *src, size_t bytes, int nthreads, std::vector<std::thread> &pool) {
  const size_t per = (bytes / nthreads + 4095) / 4096 * 4096;
  for (int q = 0; q < nthreads; ++q) {
    const size_t a = (size_t)q * per;
    if (a >= bytes) break;
    const size_t len = std::min(per, bytes - a);
    pool.emplace_back([=] { memcpy(dst + a, src + a, len); });
  }
}

// H(:, b) for b in [0, n1): rows [row0_0, row0_0 + n0) of V against rows [row0_1, row0_1 + n1); H host, ld n0
static int hk_ranges_to_host(hipStream_t st, const double *d_values, int K, double t, const double *d_vectors, int ldv,
                             int row0_0, int n0, int row0_1, int n1, double *H) {
  if (n0 == 0 || n1 == 0) return FLGP_OK;
  const size_t colbytes = sizeof(double) * (size_t)n0;
  // block width: ~512 MB per block, a multiple of 64 columns where that is possible (half a GEMM tile)
  int nc = (int)std::max<size_t>(1, ((size_t)std::max(1, tuning("hk_block_mb", 512)) << 20) / colbytes);
  if (nc >= 64) nc = nc / 64 * 64;
  if (nc > n1) nc = n1;
  const int nblk = ceil_div(n1, nc);
  if (nblk <= 1 || tuning("hk_pipelined_d2h", 1) == 0) {   // small: one contraction, one copy
    DevBuf dH, work;
    FLGP_TRY(dH.alloc(colbytes * n1));
    FLGP_TRY(work.alloc(flgp_dev_hk_workspace(n0, n1, K, 0)));
    FLGP_TRY(flgp_dev_hk(st, d_values, K, t, d_vectors, ldv, nullptr, row0_0, n0, d_vectors, ldv, nullptr, row0_1, n1,
                         dH.as<double>(), n0, work.as<double>()));
    FLGP_TRY(d2h(H, dH.p, colbytes * n1, st));
    FLGP_HIP(hipStreamSynchronize(st));
    return FLGP_OK;
  }
  RingLease lease;
  PinnedRing &ring = *lease.r;
  const size_t blkbytes = colbytes * nc;
  FLGP_TRY(ring.ensure(blkbytes));
  DevBuf dH[2], work;
  FLGP_TRY(dH[0].alloc(blkbytes)); FLGP_TRY(dH[1].alloc(blkbytes));
  FLGP_TRY(work.alloc(flgp_dev_hk_workspace(n0, nc, K, 0)));
  Stream cp;
  FLGP_TRY(cp.create());
  EventSet evs;
  FLGP_TRY(evs.create());
  hipEvent_t *gemm_done = evs.e, *dma_done = evs.e + 2;
  const int nthreads = std::max(1, std::min(tuning("hk_copy_threads", 8), (int)std::thread::hardware_concurrency()));
  std::vector<std::thread> copiers[2];
  auto join = [&](int q) { for (auto &th : copiers[q]) th.join(); copiers[q].clear(); };
  int rc = FLGP_OK;
  for (int c = 0; c <= nblk + 1 && rc == FLGP_OK; ++c) {
    const int q = c & 1;
    if (c < nblk) {
      const int b0 = c * nc, w = std::min(nc, n1 - b0);
      // device buffer q was last read by the DMA of block c-2, pinned buffer q by the host copy of block c-2
      if (c >= 2) { if (hipStreamWaitEvent(st, dma_done[q], 0) != hipSuccess) rc = FLGP_ERR_HIP; }
      if (rc == FLGP_OK)
        rc = flgp_dev_hk(st, d_values, K, t, d_vectors, ldv, nullptr, row0_0, n0, d_vectors, ldv, nullptr, row0_1 + b0, w,
                         dH[q].as<double>(), n0, work.as<double>());
      if (rc == FLGP_OK && hipEventRecord(gemm_done[q], st) != hipSuccess) rc = FLGP_ERR_HIP;
      join(q);                                   // host copy of block c-2 out of pinned buffer q
      if (rc == FLGP_OK && (hipStreamWaitEvent(cp.s, gemm_done[q], 0) != hipSuccess ||
                            hipMemcpyAsync(ring.buf[q], dH[q].p, colbytes * w, hipMemcpyDeviceToHost, cp.s) != hipSuccess ||
                            hipEventRecord(dma_done[q], cp.s) != hipSuccess)) rc = FLGP_ERR_HIP;
    }
    if (c >= 1 && c - 1 < nblk && rc == FLGP_OK) {   // block c-1 has been enqueued: when it has landed, copy it out
      const int p = (c - 1) & 1, b0 = (c - 1) * nc, w = std::min(nc, n1 - b0);
      if (hipEventSynchronize(dma_done[p]) != hipSuccess) rc = FLGP_ERR_HIP;
      else parallel_copy((char *)H + colbytes * b0, (const char *)ring.buf[p], colbytes * w, nthreads, copiers[p]);
    }
  }
  join(0); join(1);
  (void)hipStreamSynchronize(cp.s);
  (void)hipStreamSynchronize(st);
  if (rc == FLGP_ERR_HIP) set_error("HIP error in the pipelined copy of H");
  return rc;
}

// A device matrix (rows x cols, column-major, ld = rows) to the caller's pageable memory with column c at H + c * ldh
// (ldh >= rows: a rank's row block of the whole H), through the pinned ring: the DMA of block c runs while a few host
// threads copy block c-1 out of its pinned buffer.  What the multi-GPU host entry uses per rank (round 4; a single
// hipMemcpy2DAsync into pageable memory staged through the runtime's bounce buffer on one thread before).
static int d2h_cols_pipelined(hipStream_t st, const double *dM, long rows, int cols, double *H, long ldh, int rings_at_least) {
  if (rows <= 0 || cols <= 0) return FLGP_OK;
  const size_t colbytes = sizeof(double) * (size_t)rows;
  int nc = (int)std::max<size_t>(1, ((size_t)std::max(1, tuning("hk_block_mb", 512)) << 19) / colbytes);   // half of the GEMM path's block: nothing to overlap with but the copies themselves
  if (nc > cols) nc = cols;
  const int nblk = ceil_div(cols, nc);
  RingLease lease(rings_at_least);
  PinnedRing &ring = *lease.r;
  FLGP_TRY(ring.ensure(colbytes * nc));
  EventSet evs;
  FLGP_TRY(evs.create());
  hipEvent_t *dma_done = evs.e;
  const int nthreads = std::max(1, std::min(tuning("hk_copy_threads", 8), (int)std::thread::hardware_concurrency()));
  std::vector<std::thread> copiers[2];
  auto join = [&](int q) { for (auto &th : copiers[q]) th.join(); copiers[q].clear(); };
  int rc = FLGP_OK;
  for (int c = 0; c <= nblk && rc == FLGP_OK; ++c) {
    const int q = c & 1;
    if (c < nblk) {
      const int b0 = c * nc, w = std::min(nc, cols - b0);
      join(q);                                   // the host copy of block c-2 has left pinned buffer q
      if (hipMemcpyAsync(ring.buf[q], dM + (size_t)b0 * rows, colbytes * w, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipEventRecord(dma_done[q], st) != hipSuccess) rc = FLGP_ERR_HIP;
    }
    if (c >= 1 && rc == FLGP_OK) {
      const int p = (c - 1) & 1, b0 = (c - 1) * nc, w = std::min(nc, cols - b0);
      if (hipEventSynchronize(dma_done[p]) != hipSuccess) { rc = FLGP_ERR_HIP; break; }
      const char *src = (const char *)ring.buf[p];
      const int per = (w + nthreads - 1) / nthreads;
      for (int tq = 0; tq < nthreads; ++tq) {
        const int c0 = tq * per, c1 = std::min(w, c0 + per);
        if (c0 >= c1) break;
        copiers[p].emplace_back([=] {
          for (int cc = c0; cc < c1; ++cc) memcpy(H + (size_t)(b0 + cc) * (size_t)ldh, src + colbytes * (size_t)cc, colbytes);
        });
      }
    }
  }
  join(0); join(1);
  (void)hipStreamSynchronize(st);
  if (rc == FLGP_ERR_HIP) set_error("HIP error in the pipelined copy of a rank's rows of H");
  return rc;
}

}  // namespace

extern "C" void flgp_release_pinned(void) {
  std::lock_guard<std::mutex> lk(g_ring_mu);
  for (PinnedRing *c : g_rings) if (!c->busy) c->drop();
}

extern "C" int flgp_knn(const double *X, int n, int d, const double *U, int s, int r, const char *distance,
                        int *ind_knn, double *dist) {
  FLGP_TRY(check_distance(distance));
  FLGP_REQUIRE(ind_knn, "KNN: null pointer");
  if (n == 0) return FLGP_OK;
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X, n, d, U, s, d, false));
  FLGP_TRY(run_knn(S, st.s, r, dist != nullptr));
  FLGP_TRY(d2h(ind_knn, S.knn_idx.p, sizeof(int) * (size_t)n * r, st.s));
  if (dist) FLGP_TRY(d2h(dist, S.knn_dist.p, sizeof(double) * (size_t)n * r, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_v_to_z(const double *v, int r, double *z) {
  FLGP_REQUIRE(v && z, "v_to_z: null pointer");
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX, "v_to_z: need 1 <= r <= %d (got %d)", FLGP_RMAX, r);
  Stream st;
  FLGP_TRY(st.create());
  DevBuf dv, dz;
  FLGP_TRY(dv.alloc(sizeof(double) * r));
  FLGP_TRY(dz.alloc(sizeof(double) * r));
  FLGP_TRY(h2d(dv.p, v, sizeof(double) * r, st.s));
  FLGP_TRY(flgp_dev_v_to_z(st.s, dv.as<double>(), r, dz.as<double>()));
  FLGP_TRY(d2h(z, dz.p, sizeof(double) * r, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_local_anchor_embedding(const double *x, int d, const double *U, int r, double *z) {
  FLGP_REQUIRE(x && U && z, "local_anchor_embedding: null pointer");
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX, "local_anchor_embedding: need 1 <= r <= %d (got %d)", FLGP_RMAX, r);
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  // one point, its r anchors in the given order: "k-NN" indices are 0..r-1
  FLGP_TRY(upload_points(S, st.s, x, 1, d, U, r, d, false));
  S.r = r;
  std::vector<int> ids(r);
  for (int a = 0; a < r; ++a) ids[a] = a;
  FLGP_TRY(S.knn_idx.alloc(sizeof(int) * r));
  FLGP_TRY(h2d(S.knn_idx.p, ids.data(), sizeof(int) * r, st.s));
  FLGP_TRY(alloc_ell(S));
  FLGP_TRY(flgp_dev_lae(st.s, S.X.as<double>(), 1, 1, d, S.Ut.as<double>(), r, r, S.knn_idx.as<int>(), 1,
                        S.ell_idx.as<int>(), S.ell_val.as<double>()));
  FLGP_TRY(d2h(z, S.ell_val.p, sizeof(double) * r, st.s));  // sorted by anchor index == input order
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_lae(const double *X, int n, int d, const double *U, int s, int r, int *csr_p, int *csr_j,
                        double *csr_x) {
  FLGP_REQUIRE(csr_j && csr_x, "LAE: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X, n, d, U, s, d, false));
  FLGP_TRY(cross_similarity(S, st.s, r, 0, -1 /* no Laplacian */, 0.0, d));
  return csr_out(S, st.s, csr_p, csr_j, csr_x);
}

extern "C" int flgp_cross_similarity_lae(const double *X, int n, int d, const double *U, int s, int ucols, int r,
                                         const char *gl, int *csr_p, int *csr_j, double *csr_x) {
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(csr_j && csr_x, "cross_similarity_lae: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  FLGP_TRY(cross_similarity(S, st.s, r, 0, glc, 0.0, ucols));
  return csr_out(S, st.s, csr_p, csr_j, csr_x);
}

extern "C" int flgp_cross_similarity_se(const double *X, int n, int d, const double *U, int s, int ucols, int r,
                                        const char *gl, double epsilon, int *csr_p, int *csr_j, double *csr_x) {
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(csr_j && csr_x, "cross_similarity_se: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  FLGP_TRY(cross_similarity(S, st.s, r, 1, glc, epsilon, ucols));
  return csr_out(S, st.s, csr_p, csr_j, csr_x);
}

extern "C" int flgp_graph_laplacian(const int *csr_j, double *csr_x, int n, int s, int r, const char *gl,
                                    const double *num_class) {
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(csr_j && csr_x && n >= 1 && s >= 1 && r >= 1, "graph_laplacian: bad arguments");
  FLGP_REQUIRE(glc != FLGP_GL_CLUSTER_NORMALIZED || num_class, "graph_laplacian: cluster-normalized needs num_class");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  S.n = n; S.s = s; S.r = r;
  FLGP_TRY(alloc_ell(S));
  FLGP_TRY(h2d(S.ell_idx.p, csr_j, sizeof(int) * (size_t)n * r, st.s));
  FLGP_TRY(h2d(S.ell_val.p, csr_x, sizeof(double) * (size_t)n * r, st.s));
  {   // the column indices become addresses in the CSC / Gram kernels
    InputCheck ck;
    FLGP_TRY(ck.begin(st.s));
    FLGP_TRY(ck.indices(st.s, S.ell_idx.as<int>(), (long)n * r, s));
    FLGP_TRY(ck.verdict(st.s, "graph_laplacian"));
  }
  if (num_class) {
    FLGP_TRY(S.num_class.alloc(sizeof(double) * (size_t)s));
    FLGP_TRY(h2d(S.num_class.p, num_class, sizeof(double) * (size_t)s, st.s));
  }
  FLGP_TRY(laplacian(S, st.s, glc, num_class ? S.num_class.as<double>() : nullptr));
  FLGP_TRY(d2h(csr_x, S.ell_val.p, sizeof(double) * (size_t)n * r, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_spectrum_from_Z(const int *csr_j, const double *csr_x, int n, int s, int r, int K, int root,
                                    double *values, double *vectors) {
  FLGP_REQUIRE(csr_j && csr_x && values && vectors && n >= 1 && s >= 1 && r >= 1, "spectrum_from_Z: bad arguments");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  S.n = n; S.s = s; S.r = r;
  FLGP_TRY(alloc_ell(S));
  FLGP_TRY(h2d(S.ell_idx.p, csr_j, sizeof(int) * (size_t)n * r, st.s));
  FLGP_TRY(h2d(S.ell_val.p, csr_x, sizeof(double) * (size_t)n * r, st.s));
  {
    InputCheck ck;
    FLGP_TRY(ck.begin(st.s));
    FLGP_TRY(ck.indices(st.s, S.ell_idx.as<int>(), (long)n * r, s));
    FLGP_TRY(ck.finite(st.s, S.ell_val.as<double>(), (long)n * r));
    FLGP_TRY(ck.verdict(st.s, "spectrum_from_Z"));
  }
  Spectrum P;
  FLGP_TRY(spectrum(S, st.s, K, root, P, nullptr));
  FLGP_TRY(d2h(values, P.values.p, sizeof(double) * (size_t)P.K, st.s));
  FLGP_TRY(d2h(vectors, P.vectors.p, sizeof(double) * (size_t)n * P.K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_hk_from_spectrum(const double *values, const double *vectors, int n, int K, double t,
                                     const int *idx0, int n0, const int *idx1, int n1, double *H) {
  FLGP_REQUIRE(values && vectors && idx0 && idx1 && H, "HK_from_spectrum: null pointer");
  FLGP_REQUIRE(n >= 1 && K >= 1 && n0 >= 0 && n1 >= 0, "HK_from_spectrum: bad shape");
  for (int a = 0; a < n0; ++a) FLGP_REQUIRE(idx0[a] >= 0 && idx0[a] < n, "HK_from_spectrum: idx0[%d]=%d out of range", a, idx0[a]);
  for (int b = 0; b < n1; ++b) FLGP_REQUIRE(idx1[b] >= 0 && idx1[b] < n, "HK_from_spectrum: idx1[%d]=%d out of range", b, idx1[b]);
  if (n0 == 0 || n1 == 0) return FLGP_OK;
  Stream st;
  FLGP_TRY(st.create());
  DevBuf dval, dvec, di0, di1, dH, work;
  const bool r0 = is_range(idx0, n0), r1 = is_range(idx1, n1);
  FLGP_TRY(dval.alloc(sizeof(double) * K));
  FLGP_TRY(dvec.alloc(sizeof(double) * (size_t)n * K));
  FLGP_TRY(dH.alloc(sizeof(double) * (size_t)n0 * n1));
  FLGP_TRY(work.alloc(flgp_dev_hk_workspace(n0, n1, K, !r0)));
  FLGP_TRY(h2d(dval.p, values, sizeof(double) * K, st.s));
  FLGP_TRY(h2d(dvec.p, vectors, sizeof(double) * (size_t)n * K, st.s));
  if (!r0) { FLGP_TRY(di0.alloc(sizeof(int) * n0)); FLGP_TRY(h2d(di0.p, idx0, sizeof(int) * n0, st.s)); }
  if (!r1) { FLGP_TRY(di1.alloc(sizeof(int) * n1)); FLGP_TRY(h2d(di1.p, idx1, sizeof(int) * n1, st.s)); }
  FLGP_TRY(flgp_dev_hk(st.s, dval.as<double>(), K, t, dvec.as<double>(), n, r0 ? nullptr : di0.as<int>(), r0 ? idx0[0] : 0,
                       n0, dvec.as<double>(), n, r1 ? nullptr : di1.as<int>(), r1 ? idx1[0] : 0, n1, dH.as<double>(), n0,
                       work.as<double>()));
  FLGP_TRY(d2h(H, dH.p, sizeof(double) * (size_t)n0 * n1, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

// ---- device-resident EigenPair
struct flgp_eigenpair {
  DevBuf values, vectors;   // K, n x K column-major
  int n = 0, K = 0, device = 0;
};

static int hk_on_device(hipStream_t st, const double *d_values, const double *d_vectors, int n, int K, double t,
                        const int *idx0, int n0, const int *idx1, int n1, double *H) {
  for (int a = 0; a < n0; ++a) FLGP_REQUIRE(idx0[a] >= 0 && idx0[a] < n, "HK_from_spectrum: idx0[%d]=%d out of range", a, idx0[a]);
  for (int b = 0; b < n1; ++b) FLGP_REQUIRE(idx1[b] >= 0 && idx1[b] < n, "HK_from_spectrum: idx1[%d]=%d out of range", b, idx1[b]);
  DevBuf di0, di1, dH, work;
  const bool r0 = is_range(idx0, n0), r1 = is_range(idx1, n1);
  if (r0 && r1 && n0 > 0 && n1 > 0)     // the callers' usual case (LinSpaced ranges, src/Spectrum.cpp:38-39): pipelined
    return hk_ranges_to_host(st, d_values, K, t, d_vectors, n, idx0[0], n0, idx1[0], n1, H);
  FLGP_TRY(dH.alloc(sizeof(double) * (size_t)n0 * n1));
  FLGP_TRY(work.alloc(flgp_dev_hk_workspace(n0, n1, K, !r0)));
  if (!r0) { FLGP_TRY(di0.alloc(sizeof(int) * n0)); FLGP_TRY(h2d(di0.p, idx0, sizeof(int) * n0, st)); }
  if (!r1) { FLGP_TRY(di1.alloc(sizeof(int) * n1)); FLGP_TRY(h2d(di1.p, idx1, sizeof(int) * n1, st)); }
  FLGP_TRY(flgp_dev_hk(st, d_values, K, t, d_vectors, n, r0 ? nullptr : di0.as<int>(), r0 ? idx0[0] : 0, n0, d_vectors, n,
                       r1 ? nullptr : di1.as<int>(), r1 ? idx1[0] : 0, n1, dH.as<double>(), n0, work.as<double>()));
  FLGP_TRY(d2h(H, dH.p, sizeof(double) * (size_t)n0 * n1, st));
  FLGP_HIP(hipStreamSynchronize(st));
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_from_host(const double *values, const double *vectors, int n, int K, flgp_eigenpair **out) {
  FLGP_REQUIRE(values && vectors && out, "eigenpair_from_host: null pointer");
  FLGP_REQUIRE(n >= 1 && K >= 1, "eigenpair_from_host: bad shape");
  *out = nullptr;
  Stream st;
  FLGP_TRY(st.create());
  std::unique_ptr<flgp_eigenpair> ep(new flgp_eigenpair());
  ep->n = n; ep->K = K;
  FLGP_HIP(hipGetDevice(&ep->device));
  FLGP_TRY(ep->values.alloc(sizeof(double) * (size_t)K));
  FLGP_TRY(ep->vectors.alloc(sizeof(double) * (size_t)n * K));
  FLGP_TRY(h2d(ep->values.p, values, sizeof(double) * (size_t)K, st.s));
  FLGP_TRY(h2d(ep->vectors.p, vectors, sizeof(double) * (size_t)n * K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  *out = ep.release();
  return FLGP_OK;
}

extern "C" int flgp_heat_kernel_spectrum_resident(const double *X_all, int n, int d, const double *U, int s, int ucols,
                                                  int r, int K, const char *kernel, const char *gl, int root,
                                                  double epsilon, flgp_eigenpair **out) {
  int se = 0;
  FLGP_TRY(parse_kernel(kernel, &se));
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(out, "heat_kernel_spectrum_resident: null pointer");
  *out = nullptr;
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X_all, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  S.want_csc = true;
  FLGP_TRY(cross_similarity(S, st.s, r, se, glc, epsilon, ucols));
  Spectrum P;
  FLGP_TRY(spectrum(S, st.s, K, root, P, nullptr));
  FLGP_HIP(hipStreamSynchronize(st.s));
  std::unique_ptr<flgp_eigenpair> ep(new flgp_eigenpair());
  ep->n = n; ep->K = P.K;
  FLGP_HIP(hipGetDevice(&ep->device));
  std::swap(ep->values.p, P.values.p);     // the buffers change owner: no copy
  std::swap(ep->vectors.p, P.vectors.p);
  *out = ep.release();
  return FLGP_OK;
}

// ---- anchors by Lloyd k-means on the device (kmeans.hip)
extern "C" int flgp_dev_kmeans_lloyd(void *stream, const double *dX, int n, int ldx, int d, int s, double *dC, int ldc,
                                     double *d_size, int iter_max, int *iters_out, double *withinss_out);
extern "C" int flgp_dev_kmeans_init(void *stream, const double *dX, int n, int ldx, int d, const int *d_rows, int s,
                                    double *dC, int ldc);

extern "C" int flgp_kmeans_lloyd(const double *X, int n, int d, int s, const int *init_rows, int nstart, int iter_max,
                                 double *U_out, int *iters_out, double *withinss_out) {
  FLGP_REQUIRE(X && init_rows && U_out, "kmeans_lloyd: null pointer");
  FLGP_REQUIRE(n >= 1 && d >= 1 && s >= 1 && s <= n && nstart >= 1 && iter_max >= 1,
               "kmeans_lloyd: need 1 <= s <= n, nstart >= 1, iter_max >= 1");
  for (long a = 0; a < (long)nstart * s; ++a)
    FLGP_REQUIRE(init_rows[a] >= 0 && init_rows[a] < n, "kmeans_lloyd: init_rows[%ld]=%d out of range", a, init_rows[a]);
  Stream st;
  FLGP_TRY(st.create());
  DevBuf dX, dC, dbest, drows;
  FLGP_TRY(dX.alloc(sizeof(double) * (size_t)n * d));
  FLGP_TRY(dC.alloc(sizeof(double) * (size_t)s * (d + 1)));
  FLGP_TRY(drows.alloc(sizeof(int) * (size_t)nstart * s));
  if (nstart > 1) FLGP_TRY(dbest.alloc(sizeof(double) * (size_t)s * (d + 1)));
  FLGP_TRY(h2d(dX.p, X, sizeof(double) * (size_t)n * d, st.s));
  FLGP_TRY(h2d(drows.p, init_rows, sizeof(int) * (size_t)nstart * s, st.s));
  double best = 0.0;
  int best_it = 0;
  for (int j = 0; j < nstart; ++j) {
    double *C = dC.as<double>();
    FLGP_TRY(flgp_dev_kmeans_init(st.s, dX.as<double>(), n, n, d, drows.as<int>() + (size_t)j * s, s, C, s));
    int it = 0;
    double wss = 0.0;
    FLGP_TRY(flgp_dev_kmeans_lloyd(st.s, dX.as<double>(), n, n, d, s, C, s, C + (size_t)s * d, iter_max, &it,
                                   (nstart > 1 || withinss_out) ? &wss : nullptr));
    if (j == 0 || wss < best) {        // stats::kmeans keeps the start with the smallest tot.withinss
      best = wss; best_it = it;
      if (nstart > 1) FLGP_HIP(hipMemcpyAsync(dbest.p, dC.p, sizeof(double) * (size_t)s * (d + 1), hipMemcpyDeviceToDevice, st.s));
    }
  }
  FLGP_TRY(d2h(U_out, nstart > 1 ? dbest.p : dC.p, sizeof(double) * (size_t)s * (d + 1), st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  if (iters_out) *iters_out = best_it;
  if (withinss_out) *withinss_out = best;
  return FLGP_OK;
}

// ---- Nystrom-extension spectrum (reference src/Fit.cpp:244-289)
extern "C" int flgp_dev_nystrom_eigenpair(void *stream, const double *dX, int n, int ldx, int d, const double *dU, int s,
                                          int ldu, double a2, int K, double *d_values, double *d_vectors, int ldv);

static int nystrom_on_device(hipStream_t st, const double *X, int n, int d, const double *U, int s, double a2, int K,
                             DevBuf &dval, DevBuf &dvec) {
  FLGP_REQUIRE(X && U, "nystrom_eigenpair: null pointer");
  FLGP_REQUIRE(n >= 1 && d >= 1 && s >= 2 && K >= 1 && K <= s, "nystrom_eigenpair: bad shape (n=%d d=%d s=%d K=%d)", n, d, s, K);
  DevBuf dX, dU;
  FLGP_TRY(dX.alloc(sizeof(double) * (size_t)n * d));
  FLGP_TRY(dU.alloc(sizeof(double) * (size_t)s * d));
  FLGP_TRY(dval.alloc(sizeof(double) * (size_t)K));
  FLGP_TRY(dvec.alloc(sizeof(double) * (size_t)n * K));
  FLGP_TRY(h2d(dX.p, X, sizeof(double) * (size_t)n * d, st));
  FLGP_TRY(h2d(dU.p, U, sizeof(double) * (size_t)s * d, st));
  return flgp_dev_nystrom_eigenpair(st, dX.as<double>(), n, n, d, dU.as<double>(), s, s, a2, K, dval.as<double>(),
                                    dvec.as<double>(), n);
}

extern "C" int flgp_nystrom_eigenpair(const double *X, int n, int d, const double *U, int s, double a2, int K,
                                      double *values, double *vectors) {
  FLGP_REQUIRE(values && vectors, "nystrom_eigenpair: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  DevBuf dval, dvec;
  FLGP_TRY(nystrom_on_device(st.s, X, n, d, U, s, a2, K, dval, dvec));
  FLGP_TRY(d2h(values, dval.p, sizeof(double) * (size_t)K, st.s));
  FLGP_TRY(d2h(vectors, dvec.p, sizeof(double) * (size_t)n * K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_nystrom_eigenpair_resident(const double *X, int n, int d, const double *U, int s, double a2, int K,
                                               flgp_eigenpair **out) {
  FLGP_REQUIRE(out, "nystrom_eigenpair_resident: null pointer");
  *out = nullptr;
  Stream st;
  FLGP_TRY(st.create());
  std::unique_ptr<flgp_eigenpair> ep(new flgp_eigenpair());
  FLGP_TRY(nystrom_on_device(st.s, X, n, d, U, s, a2, K, ep->values, ep->vectors));
  FLGP_HIP(hipStreamSynchronize(st.s));
  ep->n = n; ep->K = K;
  FLGP_HIP(hipGetDevice(&ep->device));
  *out = ep.release();
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_dims(const flgp_eigenpair *ep, int *n, int *K) {
  FLGP_REQUIRE(ep, "eigenpair_dims: null handle");
  if (n) *n = ep->n;
  if (K) *K = ep->K;
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_to_host(const flgp_eigenpair *ep, double *values, double *vectors) {
  FLGP_REQUIRE(ep, "eigenpair_to_host: null handle");
  Stream st;
  FLGP_TRY(st.create());
  if (values) FLGP_TRY(d2h(values, ep->values.p, sizeof(double) * (size_t)ep->K, st.s));
  if (vectors) FLGP_TRY(d2h(vectors, ep->vectors.p, sizeof(double) * (size_t)ep->n * ep->K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_hk_from_eigenpair(const flgp_eigenpair *ep, int K, double t, const int *idx0, int n0, const int *idx1,
                                      int n1, double *H) {
  FLGP_REQUIRE(ep && idx0 && idx1 && H, "hk_from_eigenpair: null pointer");
  FLGP_REQUIRE(K >= 1 && K <= ep->K && n0 >= 0 && n1 >= 0, "hk_from_eigenpair: need 1 <= K <= %d", ep->K);
  if (n0 == 0 || n1 == 0) return FLGP_OK;
  Stream st;
  FLGP_TRY(st.create());
  return hk_on_device(st.s, (const double *)ep->values.p, (const double *)ep->vectors.p, ep->n, K, t, idx0, n0, idx1, n1, H);
}

// V = vectors[idx, 0:K] gathered into a dense m x K block on the device (a row range is used in place)
struct GatheredV {
  DevBuf buf, didx;
  const double *V = nullptr;
  long ld = 0;
};
static int gather_v(hipStream_t st, const flgp_eigenpair *ep, int K, const int *idx, int m, GatheredV &g) {
  FLGP_REQUIRE(ep && idx, "eigenpair: null pointer");
  FLGP_REQUIRE(K >= 1 && K <= ep->K && m >= 1, "eigenpair: need 1 <= K <= %d and m >= 1", ep->K);
  for (int a = 0; a < m; ++a) FLGP_REQUIRE(idx[a] >= 0 && idx[a] < ep->n, "eigenpair: idx[%d]=%d out of range", a, idx[a]);
  if (is_range(idx, m)) {
    g.V = (const double *)ep->vectors.p + idx[0];
    g.ld = ep->n;
    return FLGP_OK;
  }
  FLGP_TRY(g.buf.alloc(sizeof(double) * (size_t)m * K));
  FLGP_TRY(g.didx.alloc(sizeof(int) * (size_t)m));
  FLGP_TRY(h2d(g.didx.p, idx, sizeof(int) * (size_t)m, st));
  FLGP_TRY(flgp_dev_gather_rows(st, (const double *)ep->vectors.p, ep->n, g.didx.as<int>(), m, K, g.buf.as<double>()));
  g.V = g.buf.as<double>();
  g.ld = m;
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_vtv(const flgp_eigenpair *ep, int K, const int *idx, int m, double *VtV) {
  FLGP_REQUIRE(VtV, "eigenpair_vtv: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  GatheredV g;
  FLGP_TRY(gather_v(st.s, ep, K, idx, m, g));
  DevBuf out, work;
  const size_t we = (size_t)128 * K * K;
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)K * K));
  FLGP_TRY(work.alloc(sizeof(double) * we));
  // (K x m)(m x K): A(i,k) = V(k,i), B(k,j) = V(k,j)
  FLGP_TRY(gemm_launch(st.s, K, K, m, 1.0, g.V, g.ld, 1, g.V, 1, g.ld, 0.0, nullptr, 0, 0, out.as<double>(), 1, K,
                       work.as<double>(), we, 0.0, nullptr));
  FLGP_TRY(d2h(VtV, out.p, sizeof(double) * (size_t)K * K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_vty(const flgp_eigenpair *ep, int K, const int *idx, int m, const double *Y, int q,
                                  double *VtY) {
  FLGP_REQUIRE(Y && VtY && q >= 1, "eigenpair_vty: bad arguments");
  Stream st;
  FLGP_TRY(st.create());
  GatheredV g;
  FLGP_TRY(gather_v(st.s, ep, K, idx, m, g));
  DevBuf dY, out, work;
  const size_t we = (size_t)64 * K * q + 1024;
  FLGP_TRY(dY.alloc(sizeof(double) * (size_t)m * q));
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)K * q));
  FLGP_TRY(work.alloc(sizeof(double) * we));
  FLGP_TRY(h2d(dY.p, Y, sizeof(double) * (size_t)m * q, st.s));
  FLGP_TRY(gemm_launch(st.s, K, q, m, 1.0, g.V, g.ld, 1, dY.as<double>(), 1, m, 0.0, nullptr, 0, 0, out.as<double>(), 1, K,
                       work.as<double>(), we, 0.0, nullptr));
  FLGP_TRY(d2h(VtY, out.p, sizeof(double) * (size_t)K * q, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_eigenpair_vc(const flgp_eigenpair *ep, int K, const int *idx, int m, const double *C, int q,
                                 double *VC) {
  FLGP_REQUIRE(C && VC && q >= 1, "eigenpair_vc: bad arguments");
  Stream st;
  FLGP_TRY(st.create());
  GatheredV g;
  FLGP_TRY(gather_v(st.s, ep, K, idx, m, g));
  DevBuf dC, out;
  FLGP_TRY(dC.alloc(sizeof(double) * (size_t)K * q));
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)m * q));
  FLGP_TRY(h2d(dC.p, C, sizeof(double) * (size_t)K * q, st.s));
  FLGP_TRY(gemm_launch(st.s, m, q, K, 1.0, g.V, 1, g.ld, dC.as<double>(), 1, K, 0.0, nullptr, 0, 0, out.as<double>(), 1, m,
                       nullptr, 0, 0.0, nullptr));
  FLGP_TRY(d2h(VC, out.p, sizeof(double) * (size_t)m * q, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

// ---- regression consumers of the resident pair: the Woodbury algebra stays on the device --------------
namespace {
struct GprCtx {
  DevBuf ls, l, flag;
  int prepare(hipStream_t st, const flgp_eigenpair *ep, int K, double t) {
    FLGP_TRY(ls.alloc(sizeof(double) * (size_t)K)); FLGP_TRY(l.alloc(sizeof(double) * (size_t)K));
    FLGP_TRY(flag.alloc(sizeof(int)));
    FLGP_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    return gpr_weights(st, (const double *)ep->values.p, K, t, ls.as<double>(), l.as<double>());
  }
  int verdict(hipStream_t st, const char *who) {   // synchronises
    int h = 0;
    FLGP_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    FLGP_HIP(hipStreamSynchronize(st));
    if (h) { set_error("%s: the system matrix is not positive definite (Cholesky pivot <= 0)", who); return FLGP_ERR_NOCONV; }
    return FLGP_OK;
  }
};
// device index array of a gather (nullptr for a contiguous range)
int upload_idx(hipStream_t st, const int *idx, int cnt, DevBuf &buf, const int **d_out, int *row0) {
  *d_out = nullptr; *row0 = 0;
  if (is_range(idx, cnt)) { *row0 = cnt ? idx[0] : 0; return FLGP_OK; }
  FLGP_TRY(buf.alloc(sizeof(int) * (size_t)cnt));
  FLGP_TRY(h2d(buf.p, idx, sizeof(int) * (size_t)cnt, st));
  *d_out = buf.as<int>();
  return FLGP_OK;
}
}  // namespace

extern "C" int flgp_eigenpair_predict_regression(const flgp_eigenpair *ep, int K, const int *idx0, int m, const int *idx1,
                                                 int mnew, const double *Y, int q, double t, double noise, double sigma,
                                                 double *Y_pred) {
  FLGP_REQUIRE(ep && idx0 && idx1 && Y && Y_pred, "predict_regression: null pointer");
  FLGP_REQUIRE(K >= 1 && K <= ep->K && m >= 1 && mnew >= 1 && q >= 1, "predict_regression: bad shape (K=%d m=%d m_new=%d q=%d)", K, m, mnew, q);
  const double c = noise + sigma;
  FLGP_REQUIRE(c > 0.0, "predict_regression: noise + sigma must be positive");
  for (int a = 0; a < m; ++a) FLGP_REQUIRE(idx0[a] >= 0 && idx0[a] < ep->n, "predict_regression: idx0[%d]=%d out of range", a, idx0[a]);
  for (int a = 0; a < mnew; ++a) FLGP_REQUIRE(idx1[a] >= 0 && idx1[a] < ep->n, "predict_regression: idx1[%d]=%d out of range", a, idx1[a]);
  Stream st;
  FLGP_TRY(st.create());
  GprCtx G;
  FLGP_TRY(G.prepare(st.s, ep, K, t));
  const double *dval = (const double *)ep->values.p, *dvec = (const double *)ep->vectors.p;
  DevBuf dY, out;
  FLGP_TRY(dY.alloc(sizeof(double) * (size_t)m * q));
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)mnew * q));
  FLGP_TRY(h2d(dY.p, Y, sizeof(double) * (size_t)m * q, st.s));
  if (m <= K) {
    // Cvv + (sigma + noise) I, Cholesky, alpha = C^-1 Y, Y_pred = Cnv alpha        (src/Predict.cpp:48-58)
    DevBuf i0, i1, C, Cnv, work;
    const int *d0, *d1; int r0, r1;
    FLGP_TRY(upload_idx(st.s, idx0, m, i0, &d0, &r0));
    FLGP_TRY(upload_idx(st.s, idx1, mnew, i1, &d1, &r1));
    FLGP_TRY(C.alloc(sizeof(double) * (size_t)m * m));
    FLGP_TRY(Cnv.alloc(sizeof(double) * (size_t)mnew * m));
    FLGP_TRY(work.alloc(flgp_dev_hk_workspace(std::max(m, mnew), m, K, 1)));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d0, r0, m, dvec, ep->n, d0, r0, m, C.as<double>(), m, work.as<double>()));
    FLGP_TRY(gpr_add_diag(st.s, C.as<double>(), m, sigma));
    FLGP_TRY(gpr_add_diag(st.s, C.as<double>(), m, noise));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d1, r1, mnew, dvec, ep->n, d0, r0, m, Cnv.as<double>(), mnew, work.as<double>()));
    FLGP_TRY(chol_solve(st.s, C.as<double>(), m, dY.as<double>(), q, G.flag.as<int>()));
    FLGP_TRY(gemm_launch(st.s, mnew, q, m, 1.0, Cnv.as<double>(), 1, mnew, dY.as<double>(), 1, m, 0.0, nullptr, 0, 0,
                         out.as<double>(), 1, mnew, nullptr, 0, 0.0, nullptr));
  } else {
    // Woodbury: Q = Ls V^T V Ls + (noise + sigma) I  (K x K)                       (src/Predict.cpp:59-74)
    GatheredV g0, g1;
    FLGP_TRY(gather_v(st.s, ep, K, idx0, m, g0));
    FLGP_TRY(gather_v(st.s, ep, K, idx1, mnew, g1));
    DevBuf VtV, VtY, Q, R, T1, work;
    const size_t we = (size_t)128 * K * K + (size_t)64 * K * q + 1024;
    FLGP_TRY(VtV.alloc(sizeof(double) * (size_t)K * K)); FLGP_TRY(Q.alloc(sizeof(double) * (size_t)K * K));
    FLGP_TRY(VtY.alloc(sizeof(double) * (size_t)K * q)); FLGP_TRY(R.alloc(sizeof(double) * (size_t)K * q));
    FLGP_TRY(T1.alloc(sizeof(double) * (size_t)K * q)); FLGP_TRY(work.alloc(sizeof(double) * we));
    FLGP_TRY(gemm_launch(st.s, K, K, m, 1.0, g0.V, g0.ld, 1, g0.V, 1, g0.ld, 0.0, nullptr, 0, 0, VtV.as<double>(), 1, K,
                         work.as<double>(), we, 0.0, nullptr));
    FLGP_TRY(gemm_launch(st.s, K, q, m, 1.0, g0.V, g0.ld, 1, dY.as<double>(), 1, m, 0.0, nullptr, 0, 0, VtY.as<double>(), 1, K,
                         work.as<double>(), we, 0.0, nullptr));
    FLGP_TRY(gpr_q(st.s, VtV.as<double>(), G.ls.as<double>(), K, c, Q.as<double>()));
    FLGP_TRY(gpr_scale(st.s, VtY.as<double>(), G.ls.as<double>(), nullptr, K, q, R.as<double>()));          // Ls V^T Y
    FLGP_TRY(chol_solve(st.s, Q.as<double>(), K, R.as<double>(), q, G.flag.as<int>()));                       // Q^-1 (.)
    FLGP_TRY(gpr_scale(st.s, R.as<double>(), G.ls.as<double>(), nullptr, K, q, R.as<double>()));            // Ls (.)
    // V^T alpha = (V^T Y - V^T V Ls Q^-1 Ls V^T Y) / (noise + sigma)
    FLGP_TRY(gemm_launch(st.s, K, q, K, 1.0, VtV.as<double>(), 1, K, R.as<double>(), 1, K, 0.0, nullptr, 0, 0, T1.as<double>(), 1, K,
                         nullptr, 0, 0.0, nullptr));
    FLGP_TRY(gpr_diff(st.s, VtY.as<double>(), T1.as<double>(), 1.0 / c, (long)K * q, T1.as<double>()));
    FLGP_TRY(gpr_scale(st.s, T1.as<double>(), G.l.as<double>(), nullptr, K, q, T1.as<double>()));           // exp(-t lam) (.)
    FLGP_TRY(gemm_launch(st.s, mnew, q, K, 1.0, g1.V, 1, g1.ld, T1.as<double>(), 1, K, 0.0, nullptr, 0, 0, out.as<double>(), 1,
                         mnew, nullptr, 0, 0.0, nullptr));
  }
  FLGP_TRY(d2h(Y_pred, out.p, sizeof(double) * (size_t)mnew * q, st.s));
  return G.verdict(st.s, "predict_regression");
}

// predict_regression_cpp with noisepar = "different" (reference src/Predict.cpp:76-110): one noise variance per training
// row, pars = (t, noise_1 .. noise_m).  m <= K: the m x m kernel matrix with sigma + noise_i on its diagonal.  m > K: with
// Z^-1 = diag(1 / (noise_i + sigma)),  Q = Ls V^T Z^-1 V Ls + I,  alpha = Z^-1 Y - Z^-1 V Ls Q^-1 Ls V^T Z^-1 Y; only
// V^T alpha = V^T Z^-1 Y - (V^T Z^-1 V) Ls Q^-1 Ls V^T Z^-1 Y is formed (K x q), never the m x q alpha.
extern "C" int flgp_eigenpair_predict_regression_different(const flgp_eigenpair *ep, int K, const int *idx0, int m,
                                                           const int *idx1, int mnew, const double *Y, int q, double t,
                                                           const double *noise, double sigma, double *Y_pred) {
  FLGP_REQUIRE(ep && idx0 && idx1 && Y && Y_pred && noise, "predict_regression: null pointer");
  FLGP_REQUIRE(K >= 1 && K <= ep->K && m >= 1 && mnew >= 1 && q >= 1, "predict_regression: bad shape (K=%d m=%d m_new=%d q=%d)", K, m, mnew, q);
  for (int a = 0; a < m; ++a) FLGP_REQUIRE(noise[a] + sigma > 0.0, "predict_regression: noise[%d] + sigma must be positive", a);
  for (int a = 0; a < m; ++a) FLGP_REQUIRE(idx0[a] >= 0 && idx0[a] < ep->n, "predict_regression: idx0[%d]=%d out of range", a, idx0[a]);
  for (int a = 0; a < mnew; ++a) FLGP_REQUIRE(idx1[a] >= 0 && idx1[a] < ep->n, "predict_regression: idx1[%d]=%d out of range", a, idx1[a]);
  Stream st;
  FLGP_TRY(st.create());
  GprCtx G;
  FLGP_TRY(G.prepare(st.s, ep, K, t));
  const double *dval = (const double *)ep->values.p, *dvec = (const double *)ep->vectors.p;
  DevBuf dY, out, dnoise;
  FLGP_TRY(dY.alloc(sizeof(double) * (size_t)m * q));
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)mnew * q));
  FLGP_TRY(dnoise.alloc(sizeof(double) * (size_t)m));
  FLGP_TRY(h2d(dY.p, Y, sizeof(double) * (size_t)m * q, st.s));
  FLGP_TRY(h2d(dnoise.p, noise, sizeof(double) * (size_t)m, st.s));
  if (m <= K) {
    // Cvv + sigma I + diag(noise), Cholesky, alpha = C^-1 Y, Y_pred = Cnv alpha       (src/Predict.cpp:78-91)
    DevBuf i0, i1, C, Cnv, work;
    const int *d0, *d1; int r0, r1;
    FLGP_TRY(upload_idx(st.s, idx0, m, i0, &d0, &r0));
    FLGP_TRY(upload_idx(st.s, idx1, mnew, i1, &d1, &r1));
    FLGP_TRY(C.alloc(sizeof(double) * (size_t)m * m));
    FLGP_TRY(Cnv.alloc(sizeof(double) * (size_t)mnew * m));
    FLGP_TRY(work.alloc(flgp_dev_hk_workspace(std::max(m, mnew), m, K, 1)));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d0, r0, m, dvec, ep->n, d0, r0, m, C.as<double>(), m, work.as<double>()));
    FLGP_TRY(gpr_add_diag(st.s, C.as<double>(), m, sigma));
    FLGP_TRY(gpr_add_diag_vec(st.s, C.as<double>(), m, dnoise.as<double>()));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d1, r1, mnew, dvec, ep->n, d0, r0, m, Cnv.as<double>(), mnew, work.as<double>()));
    FLGP_TRY(chol_solve(st.s, C.as<double>(), m, dY.as<double>(), q, G.flag.as<int>()));
    FLGP_TRY(gemm_launch(st.s, mnew, q, m, 1.0, Cnv.as<double>(), 1, mnew, dY.as<double>(), 1, m, 0.0, nullptr, 0, 0,
                         out.as<double>(), 1, mnew, nullptr, 0, 0.0, nullptr));
  } else {
    GatheredV g0, g1;
    FLGP_TRY(gather_v(st.s, ep, K, idx0, m, g0));
    FLGP_TRY(gather_v(st.s, ep, K, idx1, mnew, g1));
    DevBuf zinv, ZV, ZY, VtZV, VtZY, Q, R, T1, work;
    const size_t we = (size_t)128 * K * K + (size_t)64 * K * q + 1024;
    FLGP_TRY(zinv.alloc(sizeof(double) * (size_t)m));
    FLGP_TRY(ZV.alloc(sizeof(double) * (size_t)m * K)); FLGP_TRY(ZY.alloc(sizeof(double) * (size_t)m * q));
    FLGP_TRY(VtZV.alloc(sizeof(double) * (size_t)K * K)); FLGP_TRY(Q.alloc(sizeof(double) * (size_t)K * K));
    FLGP_TRY(VtZY.alloc(sizeof(double) * (size_t)K * q)); FLGP_TRY(R.alloc(sizeof(double) * (size_t)K * q));
    FLGP_TRY(T1.alloc(sizeof(double) * (size_t)K * q)); FLGP_TRY(work.alloc(sizeof(double) * we));
    FLGP_TRY(gpr_zinv(st.s, dnoise.as<double>(), sigma, m, zinv.as<double>()));                                  // :98-101
    FLGP_TRY(gpr_rowscale_ld(st.s, g0.V, g0.ld, zinv.as<double>(), m, K, ZV.as<double>()));                     // Z^-1 V
    FLGP_TRY(gpr_rowscale_ld(st.s, dY.as<double>(), m, zinv.as<double>(), m, q, ZY.as<double>()));              // Z^-1 Y
    FLGP_TRY(gemm_launch(st.s, K, K, m, 1.0, g0.V, g0.ld, 1, ZV.as<double>(), 1, m, 0.0, nullptr, 0, 0, VtZV.as<double>(), 1, K,
                         work.as<double>(), we, 0.0, nullptr));                                                 // V^T Z^-1 V   :102
    FLGP_TRY(gemm_launch(st.s, K, q, m, 1.0, g0.V, g0.ld, 1, ZY.as<double>(), 1, m, 0.0, nullptr, 0, 0, VtZY.as<double>(), 1, K,
                         work.as<double>(), we, 0.0, nullptr));                                                 // V^T Z^-1 Y
    FLGP_TRY(gpr_q(st.s, VtZV.as<double>(), G.ls.as<double>(), K, 1.0, Q.as<double>()));                        // :103-104
    FLGP_TRY(gpr_scale(st.s, VtZY.as<double>(), G.ls.as<double>(), nullptr, K, q, R.as<double>()));
    FLGP_TRY(chol_solve(st.s, Q.as<double>(), K, R.as<double>(), q, G.flag.as<int>()));                         // :105-106
    FLGP_TRY(gpr_scale(st.s, R.as<double>(), G.ls.as<double>(), nullptr, K, q, R.as<double>()));
    FLGP_TRY(gemm_launch(st.s, K, q, K, 1.0, VtZV.as<double>(), 1, K, R.as<double>(), 1, K, 0.0, nullptr, 0, 0, T1.as<double>(), 1, K,
                         nullptr, 0, 0.0, nullptr));
    FLGP_TRY(gpr_diff(st.s, VtZY.as<double>(), T1.as<double>(), 1.0, (long)K * q, T1.as<double>()));            // V^T alpha
    FLGP_TRY(gpr_scale(st.s, T1.as<double>(), G.l.as<double>(), nullptr, K, q, T1.as<double>()));               // exp(-t lam) (.)
    FLGP_TRY(gemm_launch(st.s, mnew, q, K, 1.0, g1.V, 1, g1.ld, T1.as<double>(), 1, K, 0.0, nullptr, 0, 0, out.as<double>(), 1,
                         mnew, nullptr, 0, 0.0, nullptr));                                                      // :108-109
  }
  FLGP_TRY(d2h(Y_pred, out.p, sizeof(double) * (size_t)mnew * q, st.s));
  return G.verdict(st.s, "predict_regression");
}

extern "C" int flgp_eigenpair_posterior_variance(const flgp_eigenpair *ep, int K, const int *idx0, int m, const int *idx1,
                                                 int mnew, double t, double var, double sigma, double *cov) {
  FLGP_REQUIRE(ep && idx0 && idx1 && cov, "posterior_variance: null pointer");
  FLGP_REQUIRE(K >= 1 && K <= ep->K && m >= 1 && mnew >= 1, "posterior_variance: bad shape (K=%d m=%d m_new=%d)", K, m, mnew);
  const double c = var + sigma;
  FLGP_REQUIRE(c > 0.0, "posterior_variance: var + sigma must be positive");
  for (int a = 0; a < m; ++a) FLGP_REQUIRE(idx0[a] >= 0 && idx0[a] < ep->n, "posterior_variance: idx0[%d]=%d out of range", a, idx0[a]);
  for (int a = 0; a < mnew; ++a) FLGP_REQUIRE(idx1[a] >= 0 && idx1[a] < ep->n, "posterior_variance: idx1[%d]=%d out of range", a, idx1[a]);
  Stream st;
  FLGP_TRY(st.create());
  GprCtx G;
  FLGP_TRY(G.prepare(st.s, ep, K, t));
  const double *dval = (const double *)ep->values.p, *dvec = (const double *)ep->vectors.p;
  GatheredV g1;
  FLGP_TRY(gather_v(st.s, ep, K, idx1, mnew, g1));
  DevBuf out;
  FLGP_TRY(out.alloc(sizeof(double) * (size_t)mnew));
  if (m <= K) {
    // K11 = C11 + (var + sigma) I; alpha = C21 K11^-1; beta = rowsum(C21 .* alpha)          (src/Utils.cpp:227-237)
    DevBuf i0, i1, C, C12, X, work;
    const int *d0, *d1; int r0, r1;
    FLGP_TRY(upload_idx(st.s, idx0, m, i0, &d0, &r0));
    FLGP_TRY(upload_idx(st.s, idx1, mnew, i1, &d1, &r1));
    FLGP_TRY(C.alloc(sizeof(double) * (size_t)m * m));
    FLGP_TRY(C12.alloc(sizeof(double) * (size_t)m * mnew)); FLGP_TRY(X.alloc(sizeof(double) * (size_t)m * mnew));
    FLGP_TRY(work.alloc(flgp_dev_hk_workspace(m, std::max(m, mnew), K, 1)));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d0, r0, m, dvec, ep->n, d0, r0, m, C.as<double>(), m, work.as<double>()));
    FLGP_TRY(gpr_add_diag(st.s, C.as<double>(), m, c));
    FLGP_TRY(flgp_dev_hk(st.s, dval, K, t, dvec, ep->n, d0, r0, m, dvec, ep->n, d1, r1, mnew, C12.as<double>(), m, work.as<double>()));
    FLGP_HIP(hipMemcpyAsync(X.p, C12.p, sizeof(double) * (size_t)m * mnew, hipMemcpyDeviceToDevice, st.s));
    FLGP_TRY(chol_solve(st.s, C.as<double>(), m, X.as<double>(), mnew, G.flag.as<int>()));
    FLGP_TRY(gpr_rowdot(st.s, C12.as<double>(), X.as<double>(), mnew, m, g1.V, g1.ld, K, G.l.as<double>(), c, out.as<double>()));
  } else {
    // alpha = 1/(var+sigma) L V1^T (V1 - V1 Ls Q^-1 Ls V1^T V1) L; beta_i = V2(i,:) alpha V2(i,:)^T  (src/Utils.cpp:238-246)
    GatheredV g0;
    FLGP_TRY(gather_v(st.s, ep, K, idx0, m, g0));
    DevBuf VtV, Q, R, T1, W, work;
    const size_t we = (size_t)128 * K * K + 1024;
    FLGP_TRY(VtV.alloc(sizeof(double) * (size_t)K * K)); FLGP_TRY(Q.alloc(sizeof(double) * (size_t)K * K));
    FLGP_TRY(R.alloc(sizeof(double) * (size_t)K * K)); FLGP_TRY(T1.alloc(sizeof(double) * (size_t)K * K));
    FLGP_TRY(W.alloc(sizeof(double) * (size_t)mnew * K)); FLGP_TRY(work.alloc(sizeof(double) * we));
    FLGP_TRY(gemm_launch(st.s, K, K, m, 1.0, g0.V, g0.ld, 1, g0.V, 1, g0.ld, 0.0, nullptr, 0, 0, VtV.as<double>(), 1, K,
                         work.as<double>(), we, 0.0, nullptr));
    FLGP_TRY(gpr_q(st.s, VtV.as<double>(), G.ls.as<double>(), K, c, Q.as<double>()));
    FLGP_TRY(gpr_scale(st.s, VtV.as<double>(), G.ls.as<double>(), nullptr, K, K, R.as<double>()));          // Ls V1^T V1
    FLGP_TRY(chol_solve(st.s, Q.as<double>(), K, R.as<double>(), K, G.flag.as<int>()));
    FLGP_TRY(gpr_scale(st.s, R.as<double>(), G.ls.as<double>(), nullptr, K, K, R.as<double>()));            // Ls Q^-1 Ls VtV
    FLGP_TRY(gemm_launch(st.s, K, K, K, 1.0, VtV.as<double>(), 1, K, R.as<double>(), 1, K, 0.0, nullptr, 0, 0, T1.as<double>(), 1, K,
                         nullptr, 0, 0.0, nullptr));
    FLGP_TRY(gpr_diff(st.s, VtV.as<double>(), T1.as<double>(), 1.0 / c, (long)K * K, T1.as<double>()));     // (VtV - ...)/(var+sigma)
    FLGP_TRY(gpr_scale(st.s, T1.as<double>(), G.l.as<double>(), G.l.as<double>(), K, K, T1.as<double>()));   // L (.) L
    FLGP_TRY(gemm_launch(st.s, mnew, K, K, 1.0, g1.V, 1, g1.ld, T1.as<double>(), 1, K, 0.0, nullptr, 0, 0, W.as<double>(), 1,
                         mnew, nullptr, 0, 0.0, nullptr));                                                  // V2 alpha
    FLGP_TRY(gpr_rowquad(st.s, g1.V, g1.ld, W.as<double>(), mnew, K, G.l.as<double>(), c, out.as<double>()));
  }
  FLGP_TRY(d2h(cov, out.p, sizeof(double) * (size_t)mnew, st.s));
  return G.verdict(st.s, "posterior_variance");
}

extern "C" void flgp_eigenpair_free(flgp_eigenpair *ep) { delete ep; }

extern "C" int flgp_heat_kernel_spectrum(const double *X_all, int n, int d, const double *U, int s, int ucols,
                                         int r, int K, const char *kernel, const char *gl, int root,
                                         double epsilon, double *values, double *vectors) {
  int se = 0;
  FLGP_TRY(parse_kernel(kernel, &se));
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(values && vectors, "heat_kernel_spectrum: null pointer");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X_all, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  S.want_csc = true;
  FLGP_TRY(cross_similarity(S, st.s, r, se, glc, epsilon, ucols));
  Spectrum P;
  FLGP_TRY(spectrum(S, st.s, K, root, P, nullptr));
  FLGP_TRY(d2h(values, P.values.p, sizeof(double) * (size_t)P.K, st.s));
  FLGP_TRY(d2h(vectors, P.vectors.p, sizeof(double) * (size_t)n * P.K, st.s));
  FLGP_HIP(hipStreamSynchronize(st.s));
  return FLGP_OK;
}

extern "C" int flgp_heat_kernel_covariance(const double *X_all, int n, int m, int d, const double *U, int s,
                                           int ucols, int r, double t, int K, const char *kernel, const char *gl,
                                           int root, double epsilon, double *H) {
  int se = 0;
  FLGP_TRY(parse_kernel(kernel, &se));
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(H && m >= 1 && m <= n, "heat_kernel_covariance: need 1 <= m <= n and H");
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  const bool verbose = tuning("e2e_verbose", 0) != 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  FLGP_TRY(upload_points(S, st.s, X_all, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  const double t1 = now();
  S.want_csc = true;
  FLGP_TRY(cross_similarity(S, st.s, r, se, glc, epsilon, ucols));
  Spectrum P;
  FLGP_TRY(spectrum(S, st.s, K, root, P, nullptr));
  if (verbose) (void)hipStreamSynchronize(st.s);
  const double t2 = now();
  // H = V[0:n] diag(exp(-t(1-values))) V[0:m]^T  (idx0 = 0..n-1, idx1 = 0..m-1, src/Spectrum.cpp:38-40)
  FLGP_TRY(hk_ranges_to_host(st.s, P.values.as<double>(), P.K, t, P.vectors.as<double>(), n, 0, n, 0, m, H));
  if (verbose)
    fprintf(stderr, "[flgp e2e] upload %.1f ms, similarity + spectrum %.1f ms, H to host %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
            (now() - t2) * 1e3);
  return FLGP_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-sharded path behind the C ABI (include/flgp_hip.h, "Row-sharded path"): the driver that
// flgp_amd/pipeline.py is in Python, with the exchanges going through an flgp_comm table instead of torch.distributed.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void count_labels_kernel(const int *__restrict__ lab, long n, int s, int *__restrict__ cnt) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int q = lab[i];
    if (q >= 0 && q < s) atomicAdd(&cnt[q], 1);      // integer counts: exact whatever the order
  }
}
__global__ void int_to_double_kernel(const int *__restrict__ in, int n, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}
}  // namespace

extern "C" int flgp_dev_gather_anchors(void *stream, const flgp_comm *comm, const double *dU_loc, int s_loc, int d,
                                       double *dU_out, int s_total) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(dU_out && s_loc >= 0 && d >= 1 && s_total >= s_loc && (dU_loc || s_loc == 0), "gather_anchors: bad arguments");
  const int world = flgp_comm_world(comm), rank = flgp_comm_rank(comm);
  if (world == 1) {
    FLGP_REQUIRE(s_total == s_loc, "gather_anchors: one rank, but s_total != s_loc");
    if (dU_out != dU_loc && s_loc) FLGP_HIP(hipMemcpyAsync(dU_out, dU_loc, sizeof(double) * (size_t)s_loc * d, hipMemcpyDeviceToDevice, st));
    return FLGP_OK;
  }
  // the counts first (as doubles through the same collective), then the blocks padded to the largest count
  DevBuf cnt_s, cnt_r;
  FLGP_TRY(cnt_s.alloc(sizeof(double)));
  FLGP_TRY(cnt_r.alloc(sizeof(double) * (size_t)world));
  const double mine = (double)s_loc;
  FLGP_HIP(hipMemcpyAsync(cnt_s.p, &mine, sizeof(double), hipMemcpyHostToDevice, st));
  FLGP_TRY(flgp_comm_all_gather(comm, cnt_s.as<double>(), cnt_r.as<double>(), 1, st));
  std::vector<double> hc((size_t)world);
  FLGP_HIP(hipMemcpyAsync(hc.data(), cnt_r.p, sizeof(double) * (size_t)world, hipMemcpyDeviceToHost, st));
  FLGP_HIP(hipStreamSynchronize(st));
  int smax = 0; long sum = 0;
  for (int q = 0; q < world; ++q) { smax = std::max(smax, (int)hc[q]); sum += (long)hc[q]; }
  FLGP_REQUIRE(sum == s_total && (int)hc[rank] == s_loc, "gather_anchors: the ranks contribute %ld anchors, s_total = %d", sum, s_total);
  if (smax == 0) return FLGP_OK;
  DevBuf pad, all;
  FLGP_TRY(pad.alloc(sizeof(double) * (size_t)smax * d));
  FLGP_TRY(all.alloc(sizeof(double) * (size_t)smax * d * world));
  FLGP_HIP(hipMemsetAsync(pad.p, 0, sizeof(double) * (size_t)smax * d, st));
  if (s_loc)
    FLGP_HIP(hipMemcpy2DAsync(pad.p, sizeof(double) * (size_t)smax, dU_loc, sizeof(double) * (size_t)s_loc, sizeof(double) * (size_t)s_loc, d,
                              hipMemcpyDeviceToDevice, st));
  FLGP_TRY(flgp_comm_all_gather(comm, pad.as<double>(), all.as<double>(), (size_t)smax * d, st));
  long off = 0;
  for (int q = 0; q < world; ++q) {
    const int c = (int)hc[q];
    if (c)
      FLGP_HIP(hipMemcpy2DAsync(dU_out + off, sizeof(double) * (size_t)s_total, all.as<double>() + (size_t)q * smax * d,
                                sizeof(double) * (size_t)smax, sizeof(double) * (size_t)c, d, hipMemcpyDeviceToDevice, st));
    off += c;
  }
  FLGP_HIP(hipStreamSynchronize(st));     // (the temporaries die here)
  return FLGP_OK;
}

extern "C" int flgp_dev_cluster_sizes(void *stream, const flgp_comm *comm, const double *dX_loc, int n_loc, int ldx, int d,
                                      const double *dU, int ldu, int s, double *d_sizes_out) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(dX_loc && dU && d_sizes_out && n_loc >= 1 && s >= 1 && ldx >= n_loc && ldu >= s, "cluster_sizes: bad arguments");
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0, "kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  const int rows = flgp_dev_anchor_rows(s);
  DevBuf Ut, uu, idx, cnt;
  FLGP_TRY(Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(idx.alloc(sizeof(int) * (size_t)n_loc));
  FLGP_TRY(cnt.alloc(sizeof(int) * (size_t)s));
  FLGP_TRY(flgp_dev_anchor_prep(st, dU, s, ldu, d, Ut.as<double>(), uu.as<double>()));
  FLGP_TRY(flgp_dev_knn(st, dX_loc, n_loc, ldx, d, Ut.as<double>(), uu.as<double>(), s, 1, idx.as<int>(), nullptr, n_loc));
  FLGP_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int) * (size_t)s, st));
  hipLaunchKernelGGL(count_labels_kernel, dim3((unsigned)std::min<long>(2048, ((long)n_loc + 255) / 256)), dim3(256), 0, st, idx.as<int>(),
                     (long)n_loc, s, cnt.as<int>());
  hipLaunchKernelGGL(int_to_double_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, cnt.as<int>(), s, d_sizes_out);
  FLGP_TRY(check_launch("count_labels_kernel"));
  FLGP_TRY(flgp_comm_all_reduce_sum(comm, d_sizes_out, (size_t)s, st));
  FLGP_HIP(hipStreamSynchronize(st));
  return FLGP_OK;
}

extern "C" int flgp_dev_heat_kernel_covariance_sharded(void *stream, const flgp_comm *comm, const double *dX_loc, int n_loc,
                                                       int ldx, int d, long n_global, long row_lo, const double *dU, int ldu,
                                                       int s, const double *d_sizes, int m, int r, double t, int K,
                                                       const char *kernel, const char *gl, int root, double epsilon,
                                                       double *dH_loc, int ldh, double *d_values_out, double *d_vectors_out,
                                                       int ldv, int *info) {
  hipStream_t st = (hipStream_t)stream;
  int se = 0;
  FLGP_TRY(parse_kernel(kernel, &se));
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(dX_loc && dU && dH_loc, "sharded covariance: null pointer");
  FLGP_REQUIRE(n_loc >= 1 && ldx >= n_loc && d >= 1 && s >= 1 && ldu >= s, "sharded covariance: bad shape n_loc=%d d=%d s=%d", n_loc, d, s);
  FLGP_REQUIRE(n_global >= n_loc && row_lo >= 0 && row_lo + n_loc <= n_global, "sharded covariance: rows [%ld, %ld) do not lie in [0, %ld)",
               row_lo, row_lo + n_loc, n_global);
  FLGP_REQUIRE(m >= 1 && (long)m <= n_global && ldh >= n_loc, "sharded covariance: need 1 <= m <= n and ldh >= n_loc");
  FLGP_REQUIRE(glc != FLGP_GL_CLUSTER_NORMALIZED || d_sizes, "gl=\"cluster-normalized\" needs the cluster sizes");
  FLGP_REQUIRE(!d_vectors_out || ldv >= n_loc, "sharded covariance: ldv < n_loc");
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0, "kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  if (K < 0) K = s;
  Sim S;
  S.n = n_loc; S.d = d; S.s = s; S.ldx = ldx; S.comm = comm; S.n_global = n_global; S.sizes = d_sizes;
  S.X.borrow(dX_loc);
  const int rows = flgp_dev_anchor_rows(s);
  FLGP_TRY(S.Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(S.uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(flgp_dev_anchor_prep(st, dU, s, ldu, d, S.Ut.as<double>(), S.uu.as<double>()));
  S.want_csc = true;
  FLGP_TRY(cross_similarity(S, st, r, se, glc, epsilon, d));
  Spectrum P;
  FLGP_TRY(spectrum(S, st, K, root, P, info));
  // exchange 4: the training block V[0:m] (m x K), zero where this rank owns no row of it, summed over the ranks
  DevBuf V1, work;
  FLGP_TRY(V1.alloc(sizeof(double) * (size_t)m * K));
  FLGP_HIP(hipMemsetAsync(V1.p, 0, sizeof(double) * (size_t)m * K, st));
  if (row_lo < m) {
    const long cnt = std::min<long>(row_lo + n_loc, (long)m) - row_lo;
    FLGP_HIP(hipMemcpy2DAsync(V1.as<double>() + row_lo, sizeof(double) * (size_t)m, P.vectors.p, sizeof(double) * (size_t)n_loc,
                              sizeof(double) * (size_t)cnt, K, hipMemcpyDeviceToDevice, st));
  }
  FLGP_TRY(flgp_comm_all_reduce_sum(comm, V1.as<double>(), (size_t)m * K, st));
  FLGP_TRY(work.alloc(flgp_dev_hk_workspace(n_loc, m, K, 0)));
  FLGP_TRY(flgp_dev_hk(st, P.values.as<double>(), K, t, P.vectors.as<double>(), n_loc, nullptr, 0, n_loc, V1.as<double>(), m, nullptr, 0,
                       m, dH_loc, ldh, work.as<double>()));
  if (d_values_out) FLGP_HIP(hipMemcpyAsync(d_values_out, P.values.p, sizeof(double) * (size_t)K, hipMemcpyDeviceToDevice, st));
  if (d_vectors_out)
    FLGP_HIP(hipMemcpy2DAsync(d_vectors_out, sizeof(double) * (size_t)ldv, P.vectors.p, sizeof(double) * (size_t)n_loc,
                              sizeof(double) * (size_t)n_loc, K, hipMemcpyDeviceToDevice, st));
  FLGP_HIP(hipStreamSynchronize(st));
  return FLGP_OK;
}

// One rank of the sharded path at the HOST boundary: this rank's rows [row_lo, row_lo + n_loc) of X_all arrive as host
// memory (column j at X_rows + j * ldx_host) and its rows of H leave the same way (column b at H_rows + b * ldh_host).
// Upload -> input check -> the ranks AGREE that all of them can go on (flgp_comm_agree: a NaN in one shard or a failed
// allocation makes every rank return, nobody waits in an exchange for a rank that has left) -> the sharded device path ->
// H through the pinned, pipelined copy.  flgp_heat_kernel_covariance_multi runs one of these per device on its own host
// thread; a one-process-per-GPU front end (bench.py under torchrun; R with Rmpi) calls it once per process.
extern "C" int flgp_heat_kernel_covariance_rank(const flgp_comm *comm, const double *X_rows, long ldx_host, int n_loc,
                                                long n_global, long row_lo, int m, int d, const double *U, int s, int ucols,
                                                int r, double t, int K, const char *kernel, const char *gl, int root,
                                                double epsilon, double *H_rows, long ldh_host, int *info) {
  const int world = comm ? comm->world : 1;
  Stream st;
  DevBuf dX, dUall, dHl;
  // ---- everything that can fail on ONE rank alone, up to the first exchange
  auto prepare = [&]() -> int {
    int se = 0;
    FLGP_TRY(parse_kernel(kernel, &se));
    const int glc = flgp_parse_gl(gl);
    if (glc < 0) return glc;
    FLGP_REQUIRE(X_rows && U && H_rows, "heat_kernel_covariance_rank: null pointer");
    FLGP_REQUIRE(n_loc >= 1 && ldx_host >= n_loc && ldh_host >= n_loc && d >= 1 && s >= 1, "heat_kernel_covariance_rank: bad shape");
    FLGP_REQUIRE(n_global >= n_loc && row_lo >= 0 && row_lo + n_loc <= n_global && m >= 1 && (long)m <= n_global,
                 "heat_kernel_covariance_rank: rows [%ld, %ld) / m = %d do not fit n = %ld", row_lo, row_lo + n_loc, m, n_global);
    FLGP_REQUIRE(ucols == d || ucols == d + 1, "U must have d or d+1 columns (d=%d, got %d)", d, ucols);
    FLGP_REQUIRE(glc != FLGP_GL_CLUSTER_NORMALIZED || ucols == d + 1,
                 "gl=\"cluster-normalized\" needs the cluster sizes in column d+1 of U (the reference reads out of bounds here)");
    FLGP_TRY(st.create());
    FLGP_TRY(dX.alloc(sizeof(double) * (size_t)n_loc * d));
    FLGP_TRY(dUall.alloc(sizeof(double) * (size_t)s * ucols));
    FLGP_TRY(dHl.alloc(sizeof(double) * (size_t)n_loc * m));
    FLGP_HIP(hipMemcpy2DAsync(dX.p, sizeof(double) * (size_t)n_loc, X_rows, sizeof(double) * (size_t)ldx_host, sizeof(double) * (size_t)n_loc, d,
                              hipMemcpyHostToDevice, st.s));
    FLGP_TRY(h2d(dUall.p, U, sizeof(double) * (size_t)s * ucols, st.s));
    InputCheck ck;
    FLGP_TRY(ck.begin(st.s));
    FLGP_TRY(ck.finite(st.s, dX.as<double>(), (long)n_loc * d));
    FLGP_TRY(ck.finite(st.s, dUall.as<double>(), (long)s * ucols));
    return ck.verdict(st.s, "points / anchors");
  };
  int rc = prepare();
  if (world > 1) {
    if (!st.s) {                                // not even a stream: this rank cannot take part in the agreement
      flgp_comm_abort(comm);
      return rc != FLGP_OK ? rc : FLGP_ERR_HIP;
    }
    rc = flgp_comm_agree(comm, rc, st.s);
  }
  if (rc != FLGP_OK) return rc;
  const double *sizes = (ucols == d + 1) ? dUall.as<double>() + (size_t)d * s : nullptr;
  FLGP_TRY(flgp_dev_heat_kernel_covariance_sharded(st.s, comm, dX.as<double>(), n_loc, n_loc, d, n_global, row_lo, dUall.as<double>(), s, s,
                                                   sizes, m, r, t, K, kernel, gl, root, epsilon, dHl.as<double>(), n_loc, nullptr, nullptr,
                                                   0, info));
  dX.release();
  return d2h_cols_pipelined(st.s, dHl.as<double>(), n_loc, m, H_rows, ldh_host, world);
}

extern "C" int flgp_heat_kernel_covariance_multi(const double *X_all, int n, int m, int d, const double *U, int s, int ucols,
                                                 int r, double t, int K, const char *kernel, const char *gl, int root,
                                                 double epsilon, int ndev, const int *devices, double *H) {
  int se = 0;
  FLGP_TRY(parse_kernel(kernel, &se));
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(X_all && U && H && devices, "heat_kernel_covariance_multi: null pointer");
  FLGP_REQUIRE(ndev >= 1 && ndev <= 16 && n >= ndev, "heat_kernel_covariance_multi: need 1 <= ndev <= 16 ranks and n >= ndev");
  FLGP_REQUIRE(m >= 1 && m <= n && d >= 1 && s >= 1, "heat_kernel_covariance_multi: bad shape");
  FLGP_REQUIRE(ucols == d || ucols == d + 1, "U must have d or d+1 columns (d=%d, got %d)", d, ucols);
  FLGP_REQUIRE(glc != FLGP_GL_CLUSTER_NORMALIZED || ucols == d + 1,
               "gl=\"cluster-normalized\" needs the cluster sizes in column d+1 of U (the reference reads out of bounds here)");
  int ndev_sys = 0;
  FLGP_HIP(hipGetDeviceCount(&ndev_sys));
  for (int a = 0; a < ndev; ++a)
    FLGP_REQUIRE(devices[a] >= 0 && devices[a] < ndev_sys, "heat_kernel_covariance_multi: device %d is not one of the %d visible devices", devices[a], ndev_sys);
  // the caller's current device is left as it was found, on every way out (the peer-enable loop used to move it)
  struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
  } guard;
  if (ndev == 1) {
    FLGP_TRY(flgp_set_device(devices[0]));
    return flgp_heat_kernel_covariance(X_all, n, m, d, U, s, ucols, r, t, K, kernel, gl, root, epsilon, H);
  }
  // transport: RCCL when every rank has a device of its own and the library loads; else the in-process backend (ranks
  // that share a device, or no RCCL) with peer access between the devices involved
  bool distinct = true;
  for (int a = 0; a < ndev; ++a)
    for (int b = a + 1; b < ndev; ++b) distinct = distinct && devices[a] != devices[b];
  std::vector<flgp_comm *> comms((size_t)ndev, nullptr);
  bool rccl = false;
  if (distinct && tuning("multi_rccl", 1)) rccl = flgp_comm_rccl_init_all(ndev, devices, comms.data()) == FLGP_OK;
  if (!rccl) {
    FLGP_TRY(flgp_comm_inproc_create(ndev, comms.data()));
    for (int a = 0; a < ndev; ++a)
      for (int b = 0; b < ndev; ++b)
        if (devices[a] != devices[b]) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) {
            for (auto c : comms) flgp_comm_destroy(c);
            set_error("heat_kernel_covariance_multi: device %d cannot read device %d's memory and RCCL is not available", devices[a], devices[b]);
            return FLGP_ERR_UNSUPPORTED;
          }
          if (hipSetDevice(devices[a]) == hipSuccess) {
            const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); }
            (void)hipGetLastError();
          }
        }
  }
  std::vector<int> rcs((size_t)ndev, FLGP_OK);
  std::vector<std::string> msgs((size_t)ndev);
  const int fail_rank = tuning("multi_test_fail_rank", -1);     // test hook: this rank leaves without a word, as after a failed hipSetDevice (exercises the abort route: its peers are waiting in the agreement)
  auto rank_main = [&](int q) -> int {
    FLGP_HIP(hipSetDevice(devices[q]));
    const long base = n / ndev, rem = n % ndev;
    const long lo = q * base + std::min<long>(q, rem), n_loc = base + (q < rem ? 1 : 0);
    if (q == fail_rank) { set_error("injected failure before the agreement (multi_test_fail_rank)"); return FLGP_ERR_HIP; }
    return flgp_heat_kernel_covariance_rank(comms[q], X_all + lo, (long)n, (int)n_loc, (long)n, lo, m, d, U, s, ucols, r, t, K, kernel, gl,
                                            root, epsilon, H + lo, (long)n, nullptr);
  };
  std::vector<std::thread> th;
  for (int q = 0; q < ndev; ++q)
    th.emplace_back([&, q]() {
      rcs[q] = rank_main(q);
      if (rcs[q] != FLGP_OK) {
        msgs[q] = flgp_last_error();
        // the others must not wait for this rank for ever.  One process holds every rank's communicator, so ALL of them are
        // aborted from here: with RCCL a peer's kernel only ends when its OWN communicator is aborted (ncclCommAbort).
        // (FLGP_ERR_PEER = the ranks have agreed to leave together: nobody is waiting.)
        if (rcs[q] != FLGP_ERR_PEER)
          for (int c = 0; c < ndev; ++c) flgp_comm_abort(comms[c]);
      }
    });
  for (auto &x : th) x.join();
  for (auto c : comms) flgp_comm_destroy(c);
  // report the rank that failed by itself before one that merely followed it out
  for (int q = 0; q < ndev; ++q)
    if (rcs[q] != FLGP_OK && rcs[q] != FLGP_ERR_PEER && msgs[q].find("aborted by") == std::string::npos) { set_error("rank %d: %s", q, msgs[q].c_str()); return rcs[q]; }
  for (int q = 0; q < ndev; ++q)
    if (rcs[q] != FLGP_OK) { set_error("rank %d: %s", q, msgs[q].c_str()); return rcs[q]; }
  return FLGP_OK;
}

extern "C" int flgp_lae_eigenmap(const double *X, int n, int d, const double *U, int s, int ucols, int r, int ndim,
                                 const char *norm, double *eigenvalues, double *eigenvectors) {
  // lae_eigenmap (src/Spectrum.cpp:17-25): spectrum with root = true, eigenvalues = 1 - values
  FLGP_REQUIRE(eigenvalues && eigenvectors && ndim >= 1, "lae_eigenmap: bad arguments");
  FLGP_TRY(flgp_heat_kernel_spectrum(X, n, d, U, s, ucols, r, ndim, "lae", norm, 1, 0.0, eigenvalues, eigenvectors));
  for (int k = 0; k < ndim; ++k) eigenvalues[k] = 1.0 - eigenvalues[k];
  return FLGP_OK;
}


// ---------------------------------------------------------------------------------------------
// SE-kernel bandwidth grid: the spectrum part of fit_se_*_gp_cpp
// (reference src/Fit.cpp:127-178, :694-743, :820-867): ONE k-NN with distances, then for every
// bandwidth a2:  Z = exp(-dist / (a2 * mean(dist)))  ->  graphLaplacian_cpp  ->  spectrum_from_Z_cpp.
// The reference runs the l (default 10) truncated SVDs one after the other through R; here the k-NN,
// the ELL/CSC pattern and the cluster sizes are shared and the l spectra run concurrently, one host
// thread + one HIP stream each (the small dense eigen-kernels of one solve occupy a few CUs only).
// values: l x K (row i = bandwidth i), vectors: l blocks of n x K column-major.
// ---------------------------------------------------------------------------------------------
// The grid on data that is already on the device (S: points, anchors, anchor panel): one k-NN with distances, one shared
// pattern, then l spectra, `max_parallel` of them at a time on their own host thread + stream.  Results go to the host
// (values / vectors non-NULL: the reference's EigenPair per bandwidth) and / or stay on the device (d_values l x K,
// d_vectors l blocks of n x K; either may be NULL -- bench.py times the ten spectra without moving 16 GB of vectors).
static int se_grid_core(Sim &S, hipStream_t st0, int r, int K, const double *a2s, int l, int glc, int root, const double *sizes,
                        double *values, double *vectors, double *d_values, double *d_vectors, double *distances_mean_out,
                        int max_parallel, int *iters_out) {
  const int n = S.n, d = S.d, s = S.s;
  FLGP_TRY(run_knn(S, st0, r, true));
  FLGP_TRY(alloc_ell(S));
  // pattern (sorted by column) and the mean distance; the weights of this first call are discarded
  FLGP_TRY(flgp_dev_se_weights_den(st0, S.knn_idx.as<int>(), S.knn_dist.as<double>(), n, n, r, 1.0, S.ell_idx.as<int>(),
                                   S.ell_val.as<double>()));
  FLGP_TRY(build_csc(S, st0));
  DevBuf dmean, mwork;
  FLGP_TRY(dmean.alloc(sizeof(double)));
  FLGP_TRY(mwork.alloc(sizeof(double) * (size_t)(((long)n * r + 4095) / 4096 + 1)));
  FLGP_TRY(flgp_dev_mean(st0, S.knn_dist.as<double>(), (long)n * r, dmean.as<double>(), mwork.as<double>()));
  double mean = 0.0;
  FLGP_TRY(d2h(&mean, dmean.p, sizeof(double), st0));
  FLGP_HIP(hipStreamSynchronize(st0));
  if (distances_mean_out) *distances_mean_out = mean;
  int dev = 0;
  FLGP_HIP(hipGetDevice(&dev));
  (void)d;

  std::vector<int> rcs(l, FLGP_OK);
  std::vector<std::string> msgs(l);
  auto work = [&](int i) -> int {
    FLGP_HIP(hipSetDevice(dev));
    Stream ws;
    FLGP_TRY(ws.create());
    Sim W;   // shares pattern / CSC with S, owns its values
    W.n = n; W.d = S.d; W.s = s; W.r = r;
    FLGP_TRY(W.ell_val.alloc(sizeof(double) * (size_t)n * r));
    DevBuf scratch_idx;
    FLGP_TRY(scratch_idx.alloc(sizeof(int) * (size_t)n * r));
    FLGP_TRY(flgp_dev_se_weights_den(ws.s, S.knn_idx.as<int>(), S.knn_dist.as<double>(), n, n, r, a2s[i] * mean,
                                     scratch_idx.as<int>(), W.ell_val.as<double>()));
    const int *eidx = S.ell_idx.as<int>();
    const int *colptr = S.colptr.as<int>(), *pos = S.pos.as<int>();
    if (glc != FLGP_GL_RW) {
      FLGP_TRY(colsum_of(W, ws.s, eidx, W.ell_val.as<double>()));
      FLGP_TRY(flgp_dev_col_scale_row_normalize(ws.s, eidx, W.ell_val.as<double>(), n, r, W.colsum.as<double>(),
                                                glc == FLGP_GL_CLUSTER_NORMALIZED ? sizes : nullptr));
    } else {
      FLGP_TRY(flgp_dev_row_normalize(ws.s, W.ell_val.as<double>(), n, r));
    }
    FLGP_TRY(colsum_of(W, ws.s, eidx, W.ell_val.as<double>()));
    FLGP_TRY(flgp_dev_col_scale(ws.s, eidx, W.ell_val.as<double>(), n, r, W.colsum.as<double>(), nullptr, 1));
    DevBuf G, eig, V, vals, vecs, ework, uwork;
    FLGP_TRY(G.alloc(sizeof(double) * (size_t)s * s));
    FLGP_TRY(flgp_dev_gram(ws.s, eidx, W.ell_val.as<double>(), n, s, r, colptr, pos, G.as<double>(), s));
    const size_t wb = flgp_dev_eig_workspace(s, K);
    FLGP_TRY(ework.alloc(wb));
    FLGP_TRY(eig.alloc(sizeof(double) * (size_t)K));
    FLGP_TRY(V.alloc(sizeof(double) * (size_t)s * K));
    int solve_info[4] = {0, 0, 0, 0};
    FLGP_TRY(flgp_dev_eig_topk(ws.s, G.as<double>(), s, s, K, 0.0, eig.as<double>(), V.as<double>(), s, ework.p, wb, solve_info));
    if (iters_out) iters_out[i] = solve_info[0];
    FLGP_TRY(flgp_dev_spectrum_usable_route(ws.s, eig.as<double>(), K, solve_info[2]));   // (a bandwidth that underflows a column of Z ends here, with the message)
    double *vals_p = d_values ? d_values + (size_t)i * K : nullptr, *vecs_p = d_vectors ? d_vectors + (size_t)i * n * K : nullptr;
    if (!vals_p) { FLGP_TRY(vals.alloc(sizeof(double) * (size_t)K)); vals_p = vals.as<double>(); }
    if (!vecs_p) { FLGP_TRY(vecs.alloc(sizeof(double) * (size_t)n * K)); vecs_p = vecs.as<double>(); }
    FLGP_TRY(uwork.alloc(flgp_dev_u_recover_workspace(s, K)));
    FLGP_TRY(flgp_dev_u_recover(ws.s, eidx, W.ell_val.as<double>(), n, r, V.as<double>(), s, s, eig.as<double>(), K,
                                std::sqrt((double)n), root, vecs_p, n, vals_p, uwork.as<double>()));
    if (values) FLGP_TRY(d2h(values + (size_t)i * K, vals_p, sizeof(double) * (size_t)K, ws.s));
    if (vectors) FLGP_TRY(d2h(vectors + (size_t)i * n * K, vecs_p, sizeof(double) * (size_t)n * K, ws.s));
    FLGP_HIP(hipStreamSynchronize(ws.s));
    return FLGP_OK;
  };
  if (max_parallel < 1) max_parallel = 1;
  for (int i0 = 0; i0 < l; i0 += max_parallel) {
    std::vector<std::thread> th;
    for (int i = i0; i < l && i < i0 + max_parallel; ++i)
      th.emplace_back([&, i]() { rcs[i] = work(i); if (rcs[i] != FLGP_OK) msgs[i] = flgp_last_error(); });
    for (auto &t : th) t.join();
  }
  for (int i = 0; i < l; ++i)
    if (rcs[i] != FLGP_OK) { set_error("bandwidth %d (a2=%g): %s", i, a2s[i], msgs[i].c_str()); return rcs[i]; }
  return FLGP_OK;
}

extern "C" int flgp_se_spectrum_grid(const double *X_all, int n, int d, const double *U, int s, int ucols, int r,
                                     int K, const double *a2s, int l, const char *gl, int root, double *values,
                                     double *vectors, double *distances_mean_out, int max_parallel) {
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(a2s && l >= 1 && values && vectors, "se_spectrum_grid: bad arguments");
  if (K < 0) K = s;
  FLGP_REQUIRE(K >= 1 && K <= s, "need 1 <= K <= s (K=%d, s=%d)", K, s);
  Stream st;
  FLGP_TRY(st.create());
  Sim S;
  FLGP_TRY(upload_points(S, st.s, X_all, n, d, U, s, ucols, glc == FLGP_GL_CLUSTER_NORMALIZED));
  const double *sizes = (ucols == d + 1) ? S.U.as<double>() + (size_t)d * s : nullptr;
  return se_grid_core(S, st.s, r, K, a2s, l, glc, root, sizes, values, vectors, nullptr, nullptr, distances_mean_out, max_parallel, nullptr);
}

// The same on device-resident points and anchors (dX n x d column-major ld ldx; dU s x d column-major ld ldu; d_sizes s or
// NULL): the K values of every bandwidth into d_values (l x K, or NULL) and the vectors into d_vectors (l blocks of n x K
// column-major, or NULL: not kept -- ten of them are 16 GB at BASELINE configs[2]).  iters_out (or NULL): outer iterations
// of each bandwidth's eigensolve.  a2s is host memory.  Synchronises.
extern "C" int flgp_dev_se_spectrum_grid(void *stream, const double *dX, int n, int ldx, int d, const double *dU, int ldu, int s,
                                         const double *d_sizes, int r, int K, const double *a2s, int l, const char *gl, int root,
                                         double *d_values, double *d_vectors, double *distances_mean_out, int max_parallel,
                                         int *iters_out) {
  hipStream_t st = (hipStream_t)stream;
  const int glc = flgp_parse_gl(gl);
  if (glc < 0) return glc;
  FLGP_REQUIRE(dX && dU && a2s && l >= 1 && n >= 1 && ldx >= n && ldu >= s, "dev_se_spectrum_grid: bad arguments");
  FLGP_REQUIRE(glc != FLGP_GL_CLUSTER_NORMALIZED || d_sizes, "gl=\"cluster-normalized\" needs the cluster sizes");
  if (K < 0) K = s;
  FLGP_REQUIRE(K >= 1 && K <= s, "need 1 <= K <= s (K=%d, s=%d)", K, s);
  const int dpad = flgp_dev_anchor_dpad(d);
  FLGP_REQUIRE(dpad > 0, "kernels are built for 1 <= d <= %d (got %d)", FLGP_DMAX, d);
  Sim S;
  S.n = n; S.d = d; S.s = s; S.ldx = ldx;
  S.X.borrow(dX);
  const int rows = flgp_dev_anchor_rows(s);
  FLGP_TRY(S.Ut.alloc(sizeof(double) * (size_t)rows * dpad));
  FLGP_TRY(S.uu.alloc(sizeof(double) * (size_t)rows));
  FLGP_TRY(flgp_dev_anchor_prep(st, dU, s, ldu, d, S.Ut.as<double>(), S.uu.as<double>()));
  return se_grid_core(S, st, r, K, a2s, l, glc, root, d_sizes, nullptr, nullptr, d_values, d_vectors, distances_mean_out, max_parallel,
                      iters_out);
}
