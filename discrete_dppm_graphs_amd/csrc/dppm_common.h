// Shared host-side helpers for libdppm_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "dppm_hip.h"

namespace dppm {

char* error_buffer();   // thread-local, 512 bytes (dppm_common.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DPPM_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DPPM_OK;
}

#define DPPM_REQUIRE(cond, ...)                                 \
  do {                                                          \
    if (!(cond)) return ::dppm::fail(DPPM_E_INVALID, __VA_ARGS__); \
  } while (0)

#define DPPM_TRY(expr)              \
  do {                              \
    int _rc = (expr);               \
    if (_rc != DPPM_OK) return _rc; \
  } while (0)

// Opt-in launch timing (bench / profiling only): when enabled, selected launches are bracketed by HIP events on
// their own stream and summed per kernel class by dppm_prof_end().  Disabled (the default) it costs one branch.
enum ProfClass { PROF_LINEAR = 0, PROF_PLANE_MM = 1, PROF_WGRAD = 2, PROF_NORM = 3,
                 PROF_WS_MLP2 = 4, PROF_WS_SUM2 = 5, PROF_WS_COLS = 6, PROF_HEAD = 7, PROF_NCLASS = 8 };
bool prof_enabled();
void prof_before(hipStream_t st);
void prof_after(hipStream_t st, int cls, double work, double bytes = 0.0);   // work: algorithmic FLOPs, bytes: algorithmic HBM bytes of the launch

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace dppm
