#include "dppm_common.h"
#include <vector>

namespace dppm {
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

namespace {
struct ProfRec { hipEvent_t e0, e1; int cls; double work, bytes; };
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
void prof_after(hipStream_t st, int cls, double work, double bytes) {
  if (!g_prof.on || g_prof.used >= g_prof.recs.size()) return;
  ProfRec& r = g_prof.recs[g_prof.used++];
  r.cls = cls;
  r.work = work;
  r.bytes = bytes;
  (void)hipEventRecord(r.e1, st);
}
}  // namespace dppm

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

extern "C" int dppm_prof_end(double* ms_per_class, double* work_per_class, double* bytes_per_class, long* launches_per_class, int nclass) {
  using namespace dppm;
  g_prof.on = false;
  if (!ms_per_class || !work_per_class || !bytes_per_class || !launches_per_class || nclass < PROF_NCLASS)
    return fail(DPPM_E_INVALID, "dppm_prof_end: need %d classes", (int)PROF_NCLASS);
  for (int c = 0; c < nclass; ++c) { ms_per_class[c] = 0; work_per_class[c] = 0; bytes_per_class[c] = 0; launches_per_class[c] = 0; }
  for (size_t i = 0; i < g_prof.used; ++i) {
    ProfRec& r = g_prof.recs[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail(DPPM_E_LAUNCH, "dppm_prof_end: event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return fail(DPPM_E_LAUNCH, "dppm_prof_end: elapsed failed");
    ms_per_class[r.cls] += ms;
    work_per_class[r.cls] += r.work;
    bytes_per_class[r.cls] += r.bytes;
    launches_per_class[r.cls] += 1;
  }
  const int dropped = (int)(g_prof.used >= g_prof.recs.size());
  g_prof.used = 0;
  return dropped ? 1 : DPPM_OK;   // 1: the record buffer filled up, totals cover only the first max_records launches
}

extern "C" const char* dppm_last_error(void) { return dppm::error_buffer(); }
extern "C" int dppm_abi_version(void) { return 3; }
