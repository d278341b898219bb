// A/B harness: exact-f32 weight-stationary kernels (dppm_wstat.h) vs the bf16-split ones (dppm_wsplit.h).
#include "dppm_common.h"
#include "dppm_gemm.h"
#include "dppm_wstat.h"
#include "dppm_wsplit.h"
#include "w3_mlp_dma.h"
#include "w3_mlp_wide.h"
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
struct Case { const char* name; double flops; double bytes; std::function<int()> f; std::vector<float> ms; };
static std::vector<Case> cases;
static void add(const char* name, double flops, double bytes, const std::function<int()>& f) {
  if (f() != 0) { printf("%-28s FAILED: %s\n", name, dppm_last_error()); return; }
  if (hipDeviceSynchronize() != hipSuccess) { printf("%-28s device error\n", name); exit(1); }
  cases.push_back(Case{name, flops, bytes, f, {}});
}
static void run_all(int rounds) {
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
    printf("%-28s median %8.4f ms %7.1f TF %6.2f TB/s   min %8.4f ms\n", c.name, med, c.flops / med / 1e9, c.bytes / med / 1e9, mn);
  }
}
static void cmp(const char* what, const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  double m = 0, mag = 0, rms = 0, ref = 0;
  for (size_t i = 0; i < n; ++i) { const double d = (double)ha[i] - hb[i]; m = fmax(m, fabs(d)); mag = fmax(mag, fabs((double)ha[i])); rms += d * d; ref += (double)ha[i] * ha[i]; }
  printf("   %-22s max|d| = %.3g  (max|ref| = %.3g, rel %.2g)  rms rel = %.2g\n", what, m, mag, m / mag, sqrt(rms / ref));
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
  const double unit = 2.0 * H * H * (double)P * B, SB = (double)S * 4;

  WsArgs m{}; m.njobs = 2; m.maskp = mask; m.B = B; m.P = P;
  WsArgs m3 = m;
  for (int k = 0; k < 2; ++k) {
    auto& j = m.job[k]; j.X = X; j.x_bs = hs; j.W1 = W[2 * k]; j.W2 = W[2 * k + 1]; j.ldw1 = H; j.ldw2 = H; j.b1 = bias[2 * k]; j.b2 = bias[2 * k + 1]; j.Y = k ? Y1 : Y0; j.y_bs = hs;
    m3.job[k] = j; m3.job[k].Y = k ? Z1 : Z0;
  }
  add("f32 mlp pair", 4 * unit, 3 * SB, [&] { return ws_mlp_forward(m, false, 0); });
  add("bf16x3 mlp pair", 4 * unit, 3 * SB, [&] { return w3_mlp_forward<3>(m3, 0); });
  cmp("mlp x3 o1", Y0, Z0, S); cmp("mlp x3 o2", Y1, Z1, S);
  { WsArgs md = m3; md.job[0].Y = Y0; md.job[1].Y = Y1;
    hipMemset(Y0, 0, S * 4); hipMemset(Y1, 0, S * 4);
    add("bf16x3 mlp pair LDS-DMA", 4 * unit, 3 * SB, [=] { return w3d_mlp_forward(md, 0); });
    cmp("dma vs x3 o1", Z0, Y0, S); cmp("dma vs x3 o2", Z1, Y1, S); }
  { WsArgs md = m3; md.job[0].Y = Y0; md.job[1].Y = Y1;
    hipMemset(Y0, 0, S * 4); hipMemset(Y1, 0, S * 4);
    add("bf16x3 mlp pair wide", 4 * unit, 3 * SB, [=] { return w3w_mlp_forward<3, 1>(md, 0); });
    cmp("wide vs x3 o1", Z0, Y0, S); cmp("wide vs x3 o2", Z1, Y1, S);
    add("bf16x3 mlp pair wide d2", 4 * unit, 3 * SB, [=] { return w3w_mlp_forward<3, 2>(md, 0); }); }
  add("bf16x3 mlp pair depth2", 4 * unit, 3 * SB, [&] { return w3_mlp_forward<3, 2>(m3, 0); });
  add("bf16 mlp pair depth1", 4 * unit, 3 * SB, [&] { return w3_mlp_forward<1, 1>(m3, 0); });
  hipMemset(Z0, 0, S * 4); hipMemset(Z1, 0, S * 4);
  add("bf16 mlp pair", 4 * unit, 3 * SB, [&] { return w3_mlp_forward<1>(m3, 0); });
  cmp("mlp bf16 o1", Y0, Z0, S); cmp("mlp bf16 o2", Y1, Z1, S);

  WsArgs l{}; l.njobs = 1; l.B = B; l.P = P;
  { auto& j = l.job[0]; j.X = X; j.x_bs = hs; j.X2 = X2; j.x2_bs = hs; j.W1 = W[4]; j.W2 = W[4] + H; j.ldw1 = 2 * H; j.ldw2 = 2 * H; j.b1 = bias[0]; j.Y = Y0; j.y_bs = hs; }
  WsArgs l3 = l; l3.job[0].Y = Z0;
  add("f32 lin<2>", 2 * unit, 3 * SB, [&] { return ws_lin(l, 2, false, 0); });
  add("bf16x3 lin<2>", 2 * unit, 3 * SB, [&] { return w3_lin<3>(l3, 2, false, 0); });
  cmp("lin<2> x3", Y0, Z0, S);
  { WsArgs la = l3; la.job[0].accumulate = 1; la.side = X2; la.side_bs = hs; la.wside = W[5]; la.ldws = 2 * H; la.maskp = mask;
    WsArgs lb = l; lb.job[0].accumulate = 1; lb.side = X2; lb.side_bs = hs; lb.wside = W[5]; lb.ldws = 2 * H; lb.maskp = mask;
    hipMemset(Y0, 0, S * 4); hipMemset(Z0, 0, S * 4);
    ws_lin(lb, 2, false, 0); w3_lin<3>(la, 2, false, 0); hipDeviceSynchronize();
    cmp("lin<2> x3 acc+side+mask", Y0, Z0, S);
    add("f32 lin<2> acc side", 2 * unit, 4 * SB, [=] { return ws_lin(lb, 2, false, 0); });
    add("bf16x3 lin<2> acc side", 2 * unit, 4 * SB, [=] { return w3_lin<3>(la, 2, false, 0); }); }
  add("bf16 lin<2>", 2 * unit, 3 * SB, [&] { return w3_lin<1>(l3, 2, false, 0); });
  cmp("lin<2> bf16", Y0, Z0, S);

  WsArgs c = l; c.act = 1; WsArgs c3 = l3; c3.act = 1;
  add("f32 lin<1> act", unit, 2 * SB, [&] { return ws_lin(c, 1, false, 0); });
  add("bf16x3 lin<1> act", unit, 2 * SB, [&] { return w3_lin<3>(c3, 1, false, 0); });
  cmp("lin<1> x3", Y0, Z0, S);
  // transposed, two jobs sharing the source, second accumulating (a backward data-gradient launch)
  WsArgs t{}; t.njobs = 2; t.B = B; t.P = P;
  { auto& j0 = t.job[0]; j0.X = X; j0.x_bs = hs; j0.W1 = W[4]; j0.ldw1 = 2 * H; j0.Y = Y0; j0.y_bs = hs;
    auto& j1 = t.job[1]; j1.X = X; j1.x_bs = hs; j1.W1 = W[4] + H; j1.ldw1 = 2 * H; j1.Y = Y1; j1.y_bs = hs; }
  WsArgs t3 = t; t3.job[0].Y = Z0; t3.job[1].Y = Z1;
  add("f32 lin<1,T> x2", 2 * unit, 3 * SB, [&] { return ws_lin(t, 1, true, 0); });
  add("bf16x3 lin<1,T> x2", 2 * unit, 3 * SB, [&] { return w3_lin<3>(t3, 1, true, 0); });
  cmp("lin<1,T> x3 job0", Y0, Z0, S); cmp("lin<1,T> x3 job1", Y1, Z1, S);
  run_all(iters);
  return 0;
}
