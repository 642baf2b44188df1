// Error reporting, device selection and tuning knobs of libflgp_hip.so.
#include "common.h"
#include <map>
#include <mutex>
#include <string>

namespace flgp {

static thread_local char g_err[512] = "";
static std::mutex g_tune_mu;
static std::map<std::string, int> g_tune;

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int tuning(const char *key, int dflt) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  auto it = g_tune.find(key);
  return it == g_tune.end() ? dflt : it->second;
}

}  // namespace flgp

using namespace flgp;

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
  return old;
}
