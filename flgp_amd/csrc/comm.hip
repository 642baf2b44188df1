// Communicators for the row-sharded path (include/flgp_hip.h, "flgp_comm").
//
// The reference is single-process and has no communication layer ("None"); the sharded path adds four
// exchanges -- all-gather of anchors, sum all-reduce of the column sums (twice), of the packed Gram partials and of the
// zero-padded training block -- and this file is where they leave the process-local world.  `flgp_comm` is a plain table
// of two callbacks on DEVICE buffers ordered with a HIP stream, so any transport can stand behind it (an R front end with
// Rmpi, say); two are built in:
//
//   * RCCL (flgp_comm_rccl_*): the collectives over xGMI.  librccl.so is opened at run time (dlopen) -- the library
//     itself does not link against it, so it loads and the single-GPU path works where RCCL is absent.  Two ways in:
//     init_all (ONE process, one communicator per device, each driven by its own host thread: what an R session can do)
//     and init_rank (one process per GPU; the caller carries the 128-byte unique id to the other ranks).
//   * in-process (flgp_comm_inproc_create): ranks are host threads of one process; a collective is a rendezvous of the
//     threads plus a device kernel that adds the ranks' buffers in RANK ORDER (every rank gets the same bits, run after
//     run) reading the peers' memory directly -- the same device, or peer devices over xGMI once peer access is enabled.
//     It needs no library, makes every collective call site testable on a one-GPU box (two "ranks" sharing the card), and
//     is the fallback of the multi-GPU host entry point when RCCL cannot be loaded.
#include "common.h"
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace flgp {

constexpr int COMM_MAX_RANKS = 16;

struct PeerPtrs { const double *p[COMM_MAX_RANKS]; };

// out[i] = p[0][i] + p[1][i] + ... in rank order
__global__ void peer_sum_kernel(PeerPtrs pp, int world, double *__restrict__ out, size_t count) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    double v = pp.p[0][i];
    for (int q = 1; q < world; ++q) v += pp.p[q][i];
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------ in-process backend
struct InprocGroup {
  int world = 0;
  int refs = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  int failed = 0;                                  // a rank reported an error: every rank leaves its collective with it
  const double *send[COMM_MAX_RANKS];
  hipEvent_t ready[COMM_MAX_RANKS], done[COMM_MAX_RANKS];
  int device[COMM_MAX_RANKS];
  // returns the group's failure flag after everybody has arrived
  int barrier(int my_fail) {
    std::unique_lock<std::mutex> lk(mu);
    if (my_fail) failed = my_fail;
    if (aborted) return aborted;
    const long gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen || aborted; });
    return aborted ? aborted : failed;
  }
  int aborted = 0;                                 // a rank left for good: nobody waits any more
  void abort_all() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = FLGP_ERR_HIP;
    cv.notify_all();
  }
};

struct InprocComm {
  InprocGroup *g = nullptr;
  int rank = 0;
  double *tmp = nullptr;       // this rank's sum buffer (its own device)
  size_t tmp_cap = 0;
  bool events = false;
};

static int inproc_prepare(InprocComm *c) {
  if (c->events) return FLGP_OK;
  int dev = 0;
  FLGP_HIP(hipGetDevice(&dev));
  c->g->device[c->rank] = dev;
  FLGP_HIP(hipEventCreateWithFlags(&c->g->ready[c->rank], hipEventDisableTiming));
  FLGP_HIP(hipEventCreateWithFlags(&c->g->done[c->rank], hipEventDisableTiming));
  c->events = true;
  return FLGP_OK;
}

static int inproc_all_reduce_body(InprocComm *c, double *d_buf, size_t count, hipStream_t st) {
  InprocGroup *g = c->g;
  int rc = inproc_prepare(c);
  if (rc == FLGP_OK && c->tmp_cap < count) {
    if (c->tmp) (void)hipFree(c->tmp);
    c->tmp = nullptr; c->tmp_cap = 0;
    if (hipMalloc((void **)&c->tmp, sizeof(double) * count) != hipSuccess) { set_error("comm: hipMalloc of %zu doubles failed", count); rc = FLGP_ERR_NOMEM; }
    else c->tmp_cap = count;
  }
  if (rc == FLGP_OK) {
    g->send[c->rank] = d_buf;
    if (hipEventRecord(g->ready[c->rank], st) != hipSuccess) rc = FLGP_ERR_HIP;
  }
  if (g->barrier(rc)) { if (!rc) set_error("comm (in-process): aborted by another rank"); return rc ? rc : FLGP_ERR_HIP; }   // every rank has posted its buffer
  PeerPtrs pp;
  for (int q = 0; q < g->world; ++q) {
    pp.p[q] = g->send[q];
    if (q != c->rank && hipStreamWaitEvent(st, g->ready[q], 0) != hipSuccess) rc = FLGP_ERR_HIP;
  }
  if (rc == FLGP_OK) {
    const int blocks = (int)std::min<size_t>(2048, (count + 255) / 256);
    hipLaunchKernelGGL(peer_sum_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, pp, g->world, c->tmp, count);
    if (hipGetLastError() != hipSuccess || hipEventRecord(g->done[c->rank], st) != hipSuccess) rc = FLGP_ERR_HIP;
  }
  if (g->barrier(rc)) { if (!rc) set_error("comm (in-process): aborted by another rank"); return rc ? rc : FLGP_ERR_HIP; }   // every rank has enqueued its sum
  for (int q = 0; q < g->world; ++q)                            // nobody overwrites a buffer a peer still reads
    if (q != c->rank && hipStreamWaitEvent(st, g->done[q], 0) != hipSuccess) rc = FLGP_ERR_HIP;
  if (rc == FLGP_OK && hipMemcpyAsync(d_buf, c->tmp, sizeof(double) * count, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = FLGP_ERR_HIP;
  if (rc == FLGP_ERR_HIP) set_error("comm (in-process): a HIP call failed in all_reduce");
  return rc;
}

static int inproc_all_reduce(void *ctx, double *d_buf, size_t count, void *stream) {
  InprocComm *c = (InprocComm *)ctx;
  if (count == 0 || c->g->world == 1) return FLGP_OK;
  return inproc_all_reduce_body(c, d_buf, count, (hipStream_t)stream);
}

static int inproc_all_gather(void *ctx, const double *d_send, double *d_recv, size_t count, void *stream) {
  InprocComm *c = (InprocComm *)ctx;
  InprocGroup *g = c->g;
  hipStream_t st = (hipStream_t)stream;
  if (count == 0) return FLGP_OK;
  if (g->world == 1) {
    if (d_recv != d_send) FLGP_HIP(hipMemcpyAsync(d_recv, d_send, sizeof(double) * count, hipMemcpyDeviceToDevice, st));
    return FLGP_OK;
  }
  int rc = inproc_prepare(c);
  if (rc == FLGP_OK) {
    g->send[c->rank] = d_send;
    if (hipEventRecord(g->ready[c->rank], st) != hipSuccess) rc = FLGP_ERR_HIP;
  }
  if (g->barrier(rc)) { if (!rc) set_error("comm (in-process): aborted by another rank"); return rc ? rc : FLGP_ERR_HIP; }
  for (int q = 0; q < g->world && rc == FLGP_OK; ++q) {
    if (q != c->rank && hipStreamWaitEvent(st, g->ready[q], 0) != hipSuccess) rc = FLGP_ERR_HIP;
    if (rc == FLGP_OK && hipMemcpyAsync(d_recv + (size_t)q * count, g->send[q], sizeof(double) * count, hipMemcpyDeviceToDevice, st) != hipSuccess)
      rc = FLGP_ERR_HIP;
  }
  if (rc == FLGP_OK && hipEventRecord(g->done[c->rank], st) != hipSuccess) rc = FLGP_ERR_HIP;
  if (g->barrier(rc)) { if (!rc) set_error("comm (in-process): aborted by another rank"); return rc ? rc : FLGP_ERR_HIP; }
  for (int q = 0; q < g->world; ++q)                            // the send buffer stays untouched until every peer has copied it
    if (q != c->rank && hipStreamWaitEvent(st, g->done[q], 0) != hipSuccess) rc = FLGP_ERR_HIP;
  if (rc == FLGP_ERR_HIP) set_error("comm (in-process): a HIP call failed in all_gather");
  return rc;
}

static void inproc_abort(void *ctx) { if (ctx) ((InprocComm *)ctx)->g->abort_all(); }

static void inproc_destroy(void *ctx) {
  InprocComm *c = (InprocComm *)ctx;
  if (!c) return;
  if (c->tmp) (void)hipFree(c->tmp);
  if (c->events) { (void)hipEventDestroy(c->g->ready[c->rank]); (void)hipEventDestroy(c->g->done[c->rank]); }
  bool last = false;
  { std::lock_guard<std::mutex> lk(c->g->mu); last = (--c->g->refs == 0); }
  if (last) delete c->g;
  delete c;
}

// ------------------------------------------------------------------------------------------ RCCL backend (dlopen)
// The few declarations of rccl.h that are used, restated so that neither the header nor the library is needed to
// build: ncclResult_t / ncclDataType_t / ncclRedOp_t are ints (ncclSuccess = 0, ncclFloat64 = 8, ncclSum = 0), the
// communicator is an opaque pointer, the unique id 128 bytes passed by value.
typedef void *rccl_comm_t;
struct rccl_uid { char internal[128]; };
struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(rccl_uid *) = nullptr;
  int (*CommInitRank)(rccl_comm_t *, int, rccl_uid, int) = nullptr;
  int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommAbort)(rccl_comm_t) = nullptr;       // optional: without it a failing rank cannot wake its peers
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static std::mutex g_rccl_mu;
static RcclApi g_rccl;

static int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return FLGP_OK;
  void *h = nullptr;
  // an RCCL that is already in the process (a host framework's own copy) is reused rather than joined by a second one
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (h) break;
  }
  if (!h)
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  if (!h) { set_error("comm (RCCL): librccl.so could not be opened: %s", dlerror()); return FLGP_ERR_UNSUPPORTED; }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (int (*)(rccl_uid *))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(rccl_comm_t *, int, rccl_uid, int))dlsym(h, "ncclCommInitRank");
  a.CommInitAll = (int (*)(rccl_comm_t *, int, const int *))dlsym(h, "ncclCommInitAll");
  a.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
  a.CommAbort = (int (*)(rccl_comm_t))dlsym(h, "ncclCommAbort");
  a.AllReduce = (int (*)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  a.AllGather = (int (*)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
  a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.AllGather) {
    set_error("comm (RCCL): librccl.so lacks one of the nccl* entry points");
    dlclose(h);
    return FLGP_ERR_UNSUPPORTED;
  }
  g_rccl = a;
  return FLGP_OK;
}

static int rccl_fail(int r, const char *what) {
  set_error("comm (RCCL): %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return FLGP_ERR_HIP;
}

struct RcclComm {
  rccl_comm_t comm = nullptr;
  std::mutex mu;
  bool aborted = false;       // ncclCommAbort has run: it frees the communicator, so no collective and no ncclCommDestroy after it
};
constexpr int RCCL_F64 = 8, RCCL_SUM = 0;      // ncclFloat64, ncclSum

static int rccl_all_reduce(void *ctx, double *d_buf, size_t count, void *stream) {
  if (count == 0) return FLGP_OK;
  RcclComm *c = (RcclComm *)ctx;
  { std::lock_guard<std::mutex> lk(c->mu); if (c->aborted) { set_error("comm (RCCL): the communicator was aborted by a failing rank"); return FLGP_ERR_HIP; } }
  const int r = g_rccl.AllReduce(d_buf, d_buf, count, RCCL_F64, RCCL_SUM, c->comm, (hipStream_t)stream);
  return r == 0 ? FLGP_OK : rccl_fail(r, "ncclAllReduce");
}
static int rccl_all_gather(void *ctx, const double *d_send, double *d_recv, size_t count, void *stream) {
  if (count == 0) return FLGP_OK;
  RcclComm *c = (RcclComm *)ctx;
  { std::lock_guard<std::mutex> lk(c->mu); if (c->aborted) { set_error("comm (RCCL): the communicator was aborted by a failing rank"); return FLGP_ERR_HIP; } }
  const int r = g_rccl.AllGather(d_send, d_recv, count, RCCL_F64, c->comm, (hipStream_t)stream);
  return r == 0 ? FLGP_OK : rccl_fail(r, "ncclAllGather");
}
// A rank that cannot join the next exchange: ncclCommAbort ends the communicator's kernels in flight (the peers' too once
// THEIR communicators are aborted -- the single-process driver aborts all of them from the failing thread) and frees it.
static void rccl_abort(void *ctx) {
  RcclComm *c = (RcclComm *)ctx;
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->aborted || !c->comm || !g_rccl.CommAbort) return;
  c->aborted = true;
  (void)g_rccl.CommAbort(c->comm);
  c->comm = nullptr;
}
static void rccl_destroy(void *ctx) {
  RcclComm *c = (RcclComm *)ctx;
  if (!c) return;
  if (c->comm && !c->aborted && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

}  // namespace flgp

using namespace flgp;

extern "C" int flgp_comm_inproc_create(int world, flgp_comm **out) {
  FLGP_REQUIRE(out && world >= 1 && world <= COMM_MAX_RANKS, "comm: need 1 <= world <= %d and an output array", COMM_MAX_RANKS);
  InprocGroup *g = new InprocGroup();
  g->world = world; g->refs = world;
  for (int r = 0; r < world; ++r) {
    InprocComm *c = new InprocComm();
    c->g = g; c->rank = r;
    flgp_comm *t = new flgp_comm();
    t->ctx = c; t->rank = r; t->world = world;
    t->all_reduce_sum = inproc_all_reduce;
    t->all_gather = inproc_all_gather;
    t->destroy = inproc_destroy;
    t->abort = inproc_abort;
    out[r] = t;
  }
  return FLGP_OK;
}

extern "C" int flgp_comm_rccl_unique_id(void *id128) {
  FLGP_REQUIRE(id128, "comm: null id buffer");
  FLGP_TRY(rccl_load());
  rccl_uid id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r != 0) return rccl_fail(r, "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return FLGP_OK;
}

extern "C" int flgp_comm_rccl_init_rank(int world, int rank, const void *id128, flgp_comm **out) {
  FLGP_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "comm: bad rank / world");
  FLGP_TRY(rccl_load());
  rccl_uid id;
  memcpy(&id, id128, sizeof(id));
  RcclComm *c = new RcclComm();
  const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != 0) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
  flgp_comm *t = new flgp_comm();
  t->ctx = c; t->rank = rank; t->world = world;
  t->all_reduce_sum = rccl_all_reduce; t->all_gather = rccl_all_gather; t->destroy = rccl_destroy;
  t->abort = g_rccl.CommAbort ? rccl_abort : nullptr;
  *out = t;
  return FLGP_OK;
}

extern "C" int flgp_comm_rccl_init_all(int ndev, const int *devices, flgp_comm **out) {
  FLGP_REQUIRE(out && devices && ndev >= 1 && ndev <= COMM_MAX_RANKS, "comm: bad device list");
  FLGP_TRY(rccl_load());
  std::vector<rccl_comm_t> comms((size_t)ndev, nullptr);
  const int r = g_rccl.CommInitAll(comms.data(), ndev, devices);
  if (r != 0) return rccl_fail(r, "ncclCommInitAll");
  for (int q = 0; q < ndev; ++q) {
    RcclComm *c = new RcclComm();
    c->comm = comms[q];
    flgp_comm *t = new flgp_comm();
    t->ctx = c; t->rank = q; t->world = ndev;
    t->all_reduce_sum = rccl_all_reduce; t->all_gather = rccl_all_gather; t->destroy = rccl_destroy;
  t->abort = g_rccl.CommAbort ? rccl_abort : nullptr;
    out[q] = t;
  }
  return FLGP_OK;
}

extern "C" void flgp_comm_destroy(flgp_comm *c) {
  if (!c) return;
  if (c->destroy) c->destroy(c->ctx);
  delete c;
}

// convenience wrappers (ctypes / the R shim call the table through these)
extern "C" int flgp_comm_all_reduce_sum(const flgp_comm *c, double *d_buf, size_t count, void *stream) {
  if (!c || c->world <= 1) return FLGP_OK;
  FLGP_REQUIRE(c->all_reduce_sum, "comm: the table has no all_reduce_sum");
  return c->all_reduce_sum(c->ctx, d_buf, count, stream);
}
extern "C" int flgp_comm_all_gather(const flgp_comm *c, const double *d_send, double *d_recv, size_t count, void *stream) {
  if (!c || c->world <= 1) {
    if (count && d_recv != d_send) FLGP_HIP(hipMemcpyAsync(d_recv, d_send, sizeof(double) * count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FLGP_OK;
  }
  FLGP_REQUIRE(c->all_gather, "comm: the table has no all_gather");
  return c->all_gather(c->ctx, d_send, d_recv, count, stream);
}
extern "C" void flgp_comm_abort(const flgp_comm *c) { if (c && c->abort) c->abort(c->ctx); }

// Make a failure collective: every rank hands in its status (FLGP_OK or an error code) and every rank learns whether ANY
// rank failed, through one all-reduce of a flag on the communicator itself -- so that a rank whose own rows are bad (NaN in
// its shard, an allocation that failed) does not leave the others waiting in the next exchange.  Returns my_status if it is
// an error, FLGP_ERR_PEER if only other ranks failed, FLGP_OK if none did.  Synchronises the stream.
extern "C" int flgp_comm_agree(const flgp_comm *c, int my_status, void *stream) {
  if (!c || c->world <= 1) return my_status;
  hipStream_t st = (hipStream_t)stream;
  DevBuf flag;
  double h[2] = {my_status != FLGP_OK ? 1.0 : 0.0, 1.0};     // [failures, ranks that answered]
  int rc = flag.alloc(sizeof(h));
  if (rc == FLGP_OK && hipMemcpyAsync(flag.p, h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) rc = FLGP_ERR_HIP;
  if (rc == FLGP_OK && hipStreamSynchronize(st) != hipSuccess) rc = FLGP_ERR_HIP;      // (h is a stack variable)
  if (rc != FLGP_OK) {                   // this rank cannot even take part: wake the others the hard way
    if (c->abort) c->abort(c->ctx);
    return my_status != FLGP_OK ? my_status : rc;
  }
  std::string mine = my_status != FLGP_OK ? flgp_last_error() : "";
  FLGP_REQUIRE(c->all_reduce_sum, "comm: the table has no all_reduce_sum");
  rc = c->all_reduce_sum(c->ctx, flag.as<double>(), 2, stream);
  if (rc == FLGP_OK && (hipMemcpyAsync(h, flag.p, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = FLGP_ERR_HIP;
  if (my_status != FLGP_OK) { set_error("%s", mine.c_str()); return my_status; }
  if (rc != FLGP_OK) return rc;
  if (h[0] > 0.0) { set_error("comm: %d of %d ranks failed before the exchange (this rank did not); all ranks leave together", (int)h[0], (int)h[1]); return FLGP_ERR_PEER; }
  return FLGP_OK;
}

extern "C" int flgp_comm_rank(const flgp_comm *c) { return c ? c->rank : 0; }
extern "C" int flgp_comm_world(const flgp_comm *c) { return c ? c->world : 1; }
