// Shared host-side helpers for libdppm_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "dppm_hip.h"
#include "dppm_hip_debug.h"

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
// arithmetic of a timed launch (the DPPM_PREC_* values; AR_NONE: no matrix products): dppm_prof_end keeps classes apart by it, so that a
// roofline is never formed from launches of two different pipes
enum ProfArith { AR_F32 = 0, AR_BF16X3 = 1, AR_BF16 = 2, AR_F32X6 = 3, AR_NONE = 4, AR_COUNT = 5 };
inline int prof_arith_of_pieces(int ns) { return ns == 6 ? AR_F32X6 : (ns == 3 ? AR_BF16X3 : (ns == 1 ? AR_BF16 : AR_F32)); }
void prof_after(hipStream_t st, int cls, int arith, double work, double bytes = 0.0);   // work: algorithmic FLOPs, bytes: algorithmic HBM bytes of the launch

// Raises the dynamic-LDS limit of `kernel` to at least `bytes` on the CURRENT device (hipFuncSetAttribute), once per (device,
// kernel, size): thread-safe, and correct for a process that drives several GPUs (the limit is a per-device property).
int lds_limit(const void* kernel, size_t bytes, const char* what);

// Experiment switches: each selects an older kernel for the same arithmetic (A/B measurements, parity tests between forms) or, for
// the DBG_* ones, a timing experiment with wrong results.  Read ONCE per process from the environment (DPPM_<NAME>; unset or "0" = off)
// into one table; afterwards only dppm_debug_set (include/dppm_hip_debug.h) changes them.  No kernel choice depends on anything else.
enum Switch { SW_DISABLE_WSTAT, SW_DISABLE_W6, SW_DISABLE_W6MLP, SW_DISABLE_LIN32, SW_DISABLE_W4, SW_W3MLP, SW_DISABLE_MLP32, SW_DISABLE_WGRAD32,
              SW_WGRAD_F32, SW_DISABLE_STRIPMM, SW_DISABLE_PLANE4, SW_DISABLE_PLANE6, SW_DISABLE_NORMFUSE, SW_DISABLE_HEAD3,
              SW_DISABLE_INLINFUSE, SW_DISABLE_SMALLGRAPH, SW_PLANE_STRIPS_ONLY, SW_DISABLE_WIDE32, SW_DISABLE_MLP32S, SW_DISABLE_SGCHAIN, SW_DBG, SW_DBG_P4, SW_DBG_W4, SW_COUNT };
int sw(Switch s);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace dppm
