// Fused Adam over a flat float32 parameter buffer (torch.optim.Adam single-tensor semantics, amsgrad off).
// HBM-bound: 4 reads + 3 writes of 4 bytes per parameter; float4 grid-stride loop.
#include "dppm_common.h"
#include <cmath>

namespace dppm {
namespace {

struct AdamArgs { float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, gscale; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g = g * a.gscale;
  if (a.wd != 0.f) g = g + a.wd * p;
  m = m + (g - m) * (1.f - a.beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (1.f - a.beta2) * g * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, AdamArgs a) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, a); adam_one(pp.y, gg.y, mm.y, vv.y, a);
    adam_one(pp.z, gg.z, mm.z, vv.z, a); adam_one(pp.w, gg.w, mm.w, vv.w, a);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], a);
}

}  // namespace
}  // namespace dppm

using namespace dppm;

extern "C" int dppm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
  DPPM_REQUIRE(param && grad && exp_avg && exp_avg_sq, "dppm_adam_step: null pointer");
  DPPM_REQUIRE(n > 0 && step >= 1, "dppm_adam_step: n=%zu step=%d", n, step);
  DPPM_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
               "dppm_adam_step: buffers must be 16-byte aligned");
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
  a.bc1 = (float)(1.0 - std::pow((double)beta1, step));
  a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, step));
  const size_t work = (n + 3) / 4;
  const int blocks = (int)(work / 256 + 1 < 2048 ? work / 256 + 1 : 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, a);
  return check_launch("dppm_adam_step");
}
