#include "dppm_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace dppm {
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

namespace {
struct ProfRec { hipEvent_t e0, e1; int cls, arith; double work, bytes; };
struct ProfState {
  bool on = false;
  std::vector<ProfRec> recs;
  size_t used = 0;
  hipEvent_t pending = nullptr;
};
ProfState g_prof;
}  // namespace

bool prof_enabled() { return g_prof.on; }
void prof_before(hipStream_t st) {
  if (!g_prof.on || g_prof.used >= g_prof.recs.size()) return;
  (void)hipEventRecord(g_prof.recs[g_prof.used].e0, st);
}
void prof_after(hipStream_t st, int cls, int arith, double work, double bytes) {
  if (!g_prof.on || g_prof.used >= g_prof.recs.size()) return;
  ProfRec& r = g_prof.recs[g_prof.used++];
  r.cls = cls;
  r.arith = (arith >= 0 && arith < AR_COUNT) ? arith : AR_NONE;
  r.work = work;
  r.bytes = bytes;
  (void)hipEventRecord(r.e1, st);
}
}  // namespace dppm

namespace dppm {
namespace {
const char* const kSwitchNames[SW_COUNT] = {"DISABLE_WSTAT", "DISABLE_W6", "DISABLE_W6MLP", "DISABLE_LIN32", "DISABLE_W4", "W3MLP", "DISABLE_MLP32",
                                            "DISABLE_WGRAD32", "WGRAD_F32", "DISABLE_STRIPMM", "DISABLE_PLANE4", "DISABLE_PLANE6", "DISABLE_NORMFUSE",
                                            "DISABLE_HEAD3", "DISABLE_INLINFUSE", "DISABLE_SMALLGRAPH", "PLANE_STRIPS_ONLY", "DISABLE_WIDE32", "DISABLE_MLP32S", "DISABLE_SGCHAIN", "DBG", "DBG_P4", "DBG_W4"};
struct Switches {
  std::atomic<int> v[SW_COUNT];
  Switches() {
    for (int i = 0; i < SW_COUNT; ++i) {
      const std::string name = std::string("DPPM_") + kSwitchNames[i];
      const char* e = getenv(name.c_str());
      int x = 0;
      if (e) { char* end = nullptr; const long l = strtol(e, &end, 0); x = (end != e) ? (int)l : 1; }   // a non-numeric value counts as "on"
      v[i].store(x);
    }
  }
};
Switches& switches() { static Switches s; return s; }   // initialised once, thread-safe (C++11 static)
struct LdsKey { int dev; const void* k; bool operator<(const LdsKey& o) const { return dev != o.dev ? dev < o.dev : k < o.k; } };
std::mutex g_lds_mu;
std::map<LdsKey, size_t> g_lds_set;
}  // namespace

int sw(Switch s) { return switches().v[s].load(std::memory_order_relaxed); }

int lds_limit(const void* kernel, size_t bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(DPPM_E_LAUNCH, "%s: hipGetDevice failed", what);
  std::lock_guard<std::mutex> lock(g_lds_mu);
  size_t& have = g_lds_set[LdsKey{dev, kernel}];
  if (bytes > have) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
      return fail(DPPM_E_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes on device %d", what, bytes, dev);
    have = bytes;
  }
  return DPPM_OK;
}
}  // namespace dppm

extern "C" int dppm_debug_set(const char* name, int value) {
  using namespace dppm;
  if (!name) return fail(DPPM_E_INVALID, "dppm_debug_set: null name");
  for (int i = 0; i < SW_COUNT; ++i)
    if (strcmp(name, kSwitchNames[i]) == 0) { switches().v[i].store(value); return DPPM_OK; }
  return fail(DPPM_E_INVALID, "dppm_debug_set: unknown switch '%s'", name);
}
extern "C" int dppm_debug_get(const char* name, int* value) {
  using namespace dppm;
  if (!name || !value) return fail(DPPM_E_INVALID, "dppm_debug_get: null argument");
  for (int i = 0; i < SW_COUNT; ++i)
    if (strcmp(name, kSwitchNames[i]) == 0) { *value = switches().v[i].load(); return DPPM_OK; }
  return fail(DPPM_E_INVALID, "dppm_debug_get: unknown switch '%s'", name);
}

extern "C" int dppm_prof_begin(int max_records) {
  using namespace dppm;
  if (max_records <= 0) return fail(DPPM_E_INVALID, "dppm_prof_begin: max_records=%d", max_records);
  while ((int)g_prof.recs.size() < max_records) {
    ProfRec r{};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess)
      return fail(DPPM_E_LAUNCH, "dppm_prof_begin: hipEventCreate failed");
    g_prof.recs.push_back(r);
  }
  g_prof.used = 0;
  g_prof.on = true;
  return DPPM_OK;
}

extern "C" int dppm_prof_end(double* ms, double* work, double* bytes, long* launches, int nclass, int narith) {
  using namespace dppm;
  g_prof.on = false;
  if (!ms || !work || !bytes || !launches || nclass < PROF_NCLASS || narith < AR_COUNT)
    return fail(DPPM_E_INVALID, "dppm_prof_end: need %d classes x %d arithmetic kinds", (int)PROF_NCLASS, (int)AR_COUNT);
  for (int c = 0; c < nclass * narith; ++c) { ms[c] = 0; work[c] = 0; bytes[c] = 0; launches[c] = 0; }
  for (size_t i = 0; i < g_prof.used; ++i) {
    ProfRec& r = g_prof.recs[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail(DPPM_E_LAUNCH, "dppm_prof_end: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return fail(DPPM_E_LAUNCH, "dppm_prof_end: elapsed failed");
    const int k = r.cls * narith + r.arith;
    ms[k] += t;
    work[k] += r.work;
    bytes[k] += r.bytes;
    launches[k] += 1;
  }
  const int dropped = (int)(g_prof.used >= g_prof.recs.size());
  g_prof.used = 0;
  return dropped ? 1 : DPPM_OK;   // 1: the record buffer filled up, totals cover only the first max_records launches
}

extern "C" const char* dppm_last_error(void) { return dppm::error_buffer(); }
extern "C" int dppm_abi_version(void) { return 5; }
