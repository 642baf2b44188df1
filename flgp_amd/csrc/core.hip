// Error reporting, device selection and tuning knobs of libflgp_hip.so.
#include "common.h"
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace flgp {

static thread_local char g_err[512] = "";
static std::mutex g_tune_mu;
static std::map<std::string, int> g_tune;
static std::atomic<int> g_tune_count{0};

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int tuning(const char *key, int dflt) {
  if (g_tune_count.load(std::memory_order_acquire) == 0) return dflt;   // nothing set: no lock, no string
  std::lock_guard<std::mutex> lk(g_tune_mu);
  auto it = g_tune.find(key);
  return it == g_tune.end() ? dflt : it->second;
}

// ---- per-kernel HIP-event timing (off by default; bench.py switches it on) ----
struct ProfRec { int name_id; hipEvent_t e0, e1; double work; };
static std::mutex g_prof_mu;
static int g_prof_on = 0;   // 0 off, 1 only the scopes named g_prof_only, 2 every scope
static const char *const g_prof_only = "hk_panel_kernel";   // the kernel bench.py's `roofline` object is about
static std::vector<std::string> g_prof_names;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int prof_begin(const char *name, hipStream_t st, double work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return -1;
  if (g_prof_on == 1 && strcmp(name, g_prof_only) != 0) return -1;
  int id = -1;
  for (size_t i = 0; i < g_prof_names.size(); ++i) if (g_prof_names[i] == name) { id = (int)i; break; }
  if (id < 0) { g_prof_names.push_back(name); id = (int)g_prof_names.size() - 1; }
  ProfRec r{id, prof_event(), prof_event(), work};
  (void)hipEventRecord(r.e0, st);
  g_prof_recs.push_back(r);
  return (int)g_prof_recs.size() - 1;
}

void prof_end(int idx, hipStream_t st) {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < (int)g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[idx].e1, st);
}

// ---- device memory cache (common.h, DevBuf) ----
static std::mutex g_pool_mu;
static std::multimap<std::pair<int, size_t>, void *> g_pool;
static size_t g_pool_bytes = 0;

void *pool_take(int dev, size_t bytes) {
  if (tuning("pool", 1) == 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool.find(std::make_pair(dev, bytes));
  if (it == g_pool.end()) return nullptr;
  void *p = it->second;
  g_pool.erase(it);
  g_pool_bytes -= bytes;
  return p;
}

bool pool_give(int dev, void *p, size_t bytes) {
  if (tuning("pool", 1) == 0) return false;
  const size_t cap = (size_t)tuning("pool_max_mb", 16384) << 20;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool_bytes + bytes > cap) return false;
  g_pool.emplace(std::make_pair(dev, bytes), p);
  g_pool_bytes += bytes;
  return true;
}

}  // namespace flgp

using namespace flgp;

extern "C" size_t flgp_dev_pool_release(void) {
  std::vector<std::pair<int, void *>> blocks;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : g_pool) blocks.emplace_back(kv.first.first, kv.second);
    bytes = g_pool_bytes;
    g_pool.clear();
    g_pool_bytes = 0;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto &b : blocks) { (void)hipSetDevice(b.first); (void)hipFree(b.second); }
  (void)hipSetDevice(cur);
  return bytes;
}

extern "C" void flgp_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
}

extern "C" void flgp_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto &r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof_recs.clear();
}

// sums over all recorded launches of `name` (call after the stream is synchronised)
extern "C" int flgp_prof_query(const char *name, int *count, double *ms, double *work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int c = 0; double t = 0.0, w = 0.0;
  for (auto &r : g_prof_recs) {
    if (g_prof_names[r.name_id] != name) continue;
    float e = 0.f;
    if (hipEventElapsedTime(&e, r.e0, r.e1) != hipSuccess) continue;
    ++c; t += e; w += r.work;
  }
  if (count) *count = c;
  if (ms) *ms = t;
  if (work) *work = w;
  return FLGP_OK;
}

// names seen so far, '\n'-separated, into buf
extern "C" int flgp_prof_names(char *buf, int buflen) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::string all;
  for (auto &n : g_prof_names) { all += n; all += '\n'; }
  if (buf && buflen > 0) { strncpy(buf, all.c_str(), buflen - 1); buf[buflen - 1] = 0; }
  return (int)all.size();
}

extern "C" const char *flgp_last_error(void) { return g_err; }
extern "C" const char *flgp_version(void) { return "flgp-hip 0.1 (gfx950)"; }

extern "C" int flgp_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

extern "C" int flgp_set_device(int device) {
  FLGP_HIP(hipSetDevice(device));
  return FLGP_OK;
}

extern "C" int flgp_parse_gl(const char *gl) {
  if (gl) {
    if (!strcmp(gl, "rw")) return FLGP_GL_RW;
    if (!strcmp(gl, "normalized")) return FLGP_GL_NORMALIZED;
    if (!strcmp(gl, "cluster-normalized")) return FLGP_GL_CLUSTER_NORMALIZED;
  }
  // reference: Rcpp::stop("Error: the type of graph Laplacian is not supported!"), src/Utils.cpp:207
  set_error("Error: the type of graph Laplacian is not supported!");
  return FLGP_ERR_UNSUPPORTED;
}

extern "C" int flgp_set_tuning(const char *key, int value) {
  if (!key) return 0;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  int old = 0;
  auto it = g_tune.find(key);
  if (it != g_tune.end()) old = it->second;
  g_tune[key] = value;
  g_tune_count.store((int)g_tune.size(), std::memory_order_release);
  return old;
}
