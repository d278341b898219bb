// A/B harness: exact-f32 per-plane contraction (dppm_planemm.h) vs the split-bf16 one (dppm_planesplit.h).
#include "dppm_common.h"
#include "dppm_gemm.h"
#include "dppm_planemm.h"
#include "dppm_planesplit.h"
#include <vector>
#include <cstdio>
#include <cmath>
#include <functional>
#include <algorithm>
using namespace dppm;
static float* dalloc(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
  float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice); return d;
}
static void cmp(const char* what, const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  double m = 0, mag = 0, rms = 0, ref = 0; size_t bad = 0;
  for (size_t i = 0; i < n; ++i) { const double d = (double)ha[i] - hb[i]; if (!(fabs(d) < 1e30)) ++bad; m = fmax(m, fabs(d)); mag = fmax(mag, fabs((double)ha[i])); rms += d * d; ref += (double)ha[i] * ha[i]; }
  printf("   %-22s max|d| = %.3g  (max|ref| = %.3g, rel %.2g)  rms rel = %.2g  nonfinite %zu\n", what, m, mag, m / mag, sqrt(rms / ref), bad);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  for (int n : {200, 129, 256, 180}) {
    const int ld = (n + 3) / 4 * 4, batch = n == 200 ? 4096 : 512;
    const size_t P = (size_t)n * ld, S = P * batch;
    float* A = dalloc(S, 1.f, 1); float* B = dalloc(S, 1.f, 2);
    float *C0, *C1; hipMalloc(&C0, S * 4); hipMalloc(&C1, S * 4);
    hipMemset(C0, 0xff, S * 4); hipMemset(C1, 0xff, S * 4);
    float* scale = dalloc(batch / 128 + 1, 0.1f, 3);
    PlaneArgs a0{A, B, C0, n, ld, scale, nullptr, 128};
    PlaneArgs a1{A, B, C1, n, ld, scale, nullptr, 128};
    if (plane_mm_launch(a0, batch, false, true, 0)) { printf("f32 failed: %s\n", dppm_last_error()); return 1; }
    if (plane3_launch<3>(a1, batch, 0)) { printf("x3 failed: %s\n", dppm_last_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    printf("n = %d\n", n);
    cmp("x3 vs f32", C0, C1, S);
    if (plane3_launch<1>(a1, batch, 0)) { printf("bf16 failed: %s\n", dppm_last_error()); return 1; }
    hipDeviceSynchronize();
    cmp("bf16 vs f32", C0, C1, S);
    if (n == 200) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      std::vector<float> t0, t1, t2;
      for (int r = 0; r < iters; ++r) {
        float ms;
        hipEventRecord(e0, 0); for (int i = 0; i < 3; ++i) plane_mm_launch(a0, batch, false, true, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); t0.push_back(ms / 3);
        hipEventRecord(e0, 0); for (int i = 0; i < 3; ++i) plane3_launch<3>(a1, batch, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); t1.push_back(ms / 3);
        hipEventRecord(e0, 0); for (int i = 0; i < 3; ++i) plane3_launch<1>(a1, batch, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); t2.push_back(ms / 3);
      }
      std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end()); std::sort(t2.begin(), t2.end());
      const double fl = 2.0 * n * n * n * batch, by = 3.0 * S * 4;
      printf("f32    median %.4f ms %.1f TF %.2f TB/s\n", t0[iters / 2], fl / t0[iters / 2] / 1e9, by / t0[iters / 2] / 1e9);
      printf("bf16x3 median %.4f ms %.1f TF %.2f TB/s\n", t1[iters / 2], fl / t1[iters / 2] / 1e9, by / t1[iters / 2] / 1e9);
      printf("bf16   median %.4f ms %.1f TF %.2f TB/s\n", t2[iters / 2], fl / t2[iters / 2] / 1e9, by / t2[iters / 2] / 1e9);
    }
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(scale);
  }
  return 0;
}
