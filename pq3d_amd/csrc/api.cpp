// Host-side plumbing of libpq3d_hip.so: version + last-error text (thread-safe enough: one string per thread).
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/pq3d_hip.h"

static thread_local std::string g_err;

extern "C" void pq3d_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* pq3d_last_error(void) { return g_err.c_str(); }
extern "C" int pq3d_version(void) { return 1; }

// number of devices this process can see (cached): single-device processes skip the per-call device guard (common.h)
int pq3d_visible_devices() {
  static const int n = [] { int c = 0; return hipGetDeviceCount(&c) == hipSuccess ? c : 1; }();
  return n;
}
