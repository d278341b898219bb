// Stand-alone A/B harness for the weight-stationary kernels (bench geometry B=32, P=40000, H=128).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../discrete_dppm_graphs_amd/csrc bench.hip ../../discrete_dppm_graphs_amd/csrc/dppm_common.hip -o wsbench
#include "dppm_common.h"
#include "dppm_gemm.h"
#include "wstat_v2.h"
#include "dppm_wstat.h"
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
struct Case { const char* name; double flops; std::function<int()> f; std::vector<float> ms; };
static std::vector<Case> cases;
static double timeit(const char* name, double flops, int, const std::function<int()>& f) {
  if (f() != 0) { printf("%-28s FAILED: %s\n", name, dppm_last_error()); return 0; }
  hipDeviceSynchronize();
  cases.push_back(Case{name, flops, f, {}});
  return 0;
}
static void run_all(int rounds) {   // interleaved: every round runs each case 3 times back to back
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < rounds; ++r)
    for (auto& c : cases) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 3; ++i) c.f();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); c.ms.push_back(ms / 3);
    }
  for (auto& c : cases) {
    std::sort(c.ms.begin(), c.ms.end());
    const float med = c.ms[c.ms.size() / 2], mn = c.ms[0];
    printf("%-28s median %8.4f ms %7.1f TF   min %8.4f ms %7.1f TF\n", c.name, med, c.flops / med / 1e9, mn, c.flops / mn / 1e9);
  }
}
static double maxdiff(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  double m = 0; for (size_t i = 0; i < n; ++i) m = fmax(m, fabs((double)ha[i] - hb[i])); return m;
}

int main(int argc, char** argv) {
  const int B = 32, H = 128, P = 40000, iters = argc > 1 ? atoi(argv[1]) : 10;
  const size_t S = (size_t)B * H * P;
  float* X = dalloc(S, 1.f, 1); float* X2 = dalloc(S, 1.f, 2);
  float* W[6]; for (int i = 0; i < 6; ++i) W[i] = dalloc((size_t)H * 2 * H, 0.09f, 10 + i);
  float* bias[4]; for (int i = 0; i < 4; ++i) bias[i] = dalloc(H, 0.1f, 20 + i);
  std::vector<float> hm((size_t)B * P); for (size_t i = 0; i < hm.size(); ++i) hm[i] = (i % 7 == 3) ? 0.f : 1.f;
  float* mask; hipMalloc(&mask, hm.size() * 4); hipMemcpy(mask, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
  float *Y0, *Y1, *Z0, *Z1; hipMalloc(&Y0, S * 4); hipMalloc(&Y1, S * 4); hipMalloc(&Z0, S * 4); hipMalloc(&Z1, S * 4);
  const long hs = (long)H * P;
  const double unit = 2.0 * H * H * (double)P * B;

  // ---- MLP pair forward
  v2::WsArgs a2{}; a2.njobs = 2; a2.maskp = mask; a2.B = B; a2.P = P;
  WsArgs a3{}; a3.njobs = 2; a3.maskp = mask; a3.B = B; a3.P = P;
  for (int k = 0; k < 2; ++k) {
    auto& j2 = a2.job[k]; j2.X = X; j2.x_bs = hs; j2.W1 = W[2 * k]; j2.W2 = W[2 * k + 1]; j2.ldw1 = H; j2.ldw2 = H; j2.b1 = bias[2 * k]; j2.b2 = bias[2 * k + 1]; j2.Y = k ? Y1 : Y0; j2.y_bs = hs;
    auto& j3 = a3.job[k]; j3.X = X; j3.x_bs = hs; j3.W1 = W[2 * k]; j3.W2 = W[2 * k + 1]; j3.ldw1 = H; j3.ldw2 = H; j3.b1 = bias[2 * k]; j3.b2 = bias[2 * k + 1]; j3.Y = k ? Z1 : Z0; j3.y_bs = hs;
  }
  timeit("v2 mlp pair", 4 * unit, iters, [&] { return v2::ws_mlp_forward(a2, false, 0); });
  timeit("new mlp pair", 4 * unit, iters, [&] { return ws_mlp_forward(a3, false, 0); });
  printf("   max |v2 - v3| = %.3g %.3g\n", maxdiff(Y0, Z0, S), maxdiff(Y1, Z1, S));

  // ---- two-source linear (m4)
  v2::WsArgs b2{}; b2.njobs = 1; b2.B = B; b2.P = P;
  { auto& j = b2.job[0]; j.X = X; j.x_bs = hs; j.X2 = X2; j.x2_bs = hs; j.W1 = W[4]; j.W2 = W[4] + H; j.ldw1 = 2 * H; j.ldw2 = 2 * H; j.b1 = bias[0]; j.Y = Y0; j.y_bs = hs; }
  WsArgs b3{}; b3.njobs = 1; b3.B = B; b3.P = P;
  { auto& j = b3.job[0]; j.X = X; j.x_bs = hs; j.X2 = X2; j.x2_bs = hs; j.W1 = W[4]; j.W2 = W[4] + H; j.ldw1 = 2 * H; j.ldw2 = 2 * H; j.b1 = bias[0]; j.Y = Z0; j.y_bs = hs; }
  timeit("v2 lin<2>", 2 * unit, iters, [&] { return v2::ws_lin(b2, 2, false, 0); });
  timeit("new lin<2>", 2 * unit, iters, [&] { return ws_lin(b3, 2, false, 0); });
  printf("   max |v2 - v3| = %.3g\n", maxdiff(Y0, Z0, S));

  // ---- one-source linear with LeakyReLU (after_conv)
  v2::WsArgs c2 = b2; c2.act = 1;
  WsArgs c3 = b3; c3.act = 1;
  timeit("v2 lin<1>", unit, iters, [&] { return v2::ws_lin(c2, 1, false, 0); });
  timeit("new lin<1>", unit, iters, [&] { return ws_lin(c3, 1, false, 0); });
  printf("   max |v2 - v3| = %.3g\n", maxdiff(Y0, Z0, S));
  run_all(iters);
  return 0;
}
