// Host-side plumbing of libpq3d_hip.so: version + last-error text (thread-safe enough: one string per thread).
#include <string>

#include "../../include/pq3d_hip.h"

static thread_local std::string g_err;

extern "C" void pq3d_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* pq3d_last_error(void) { return g_err.c_str(); }
extern "C" int pq3d_version(void) { return 1; }
