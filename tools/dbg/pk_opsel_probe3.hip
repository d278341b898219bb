// Third probe for the packed-f32 operand-select effect, shaped after the build that fails most often: the norm's
// x * scale + shift in head3_kernel's stagers as v_pk_fma_f32 with the HIGH register of an LDS-read (scale, shift) pair broadcast into
// both halves (op_sel:[0,1,1]), applied in place to the destination registers of global loads that stream from HBM (a 512 MB source:
// the earlier probes read an L2-resident array), three tiles in flight per thread, while eight other waves of the workgroup read LDS
// fragments and run MFMAs.  MODE 0 writes the packed form's results, MODE 1 the same arithmetic as two v_fma_f32, MODE 2 the packed form
// with the LOW register broadcast (op_sel_hi:[1,0,0]; compared with its own scalar twin, MODE 3); the host compares the outputs.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/dbg/pk_opsel_probe3 tools/dbg/pk_opsel_probe3.hip     run: tools/dbg/pk_opsel_probe3 [repeats]
#ifndef PARTNER
#define PARTNER 2   // what the eight other waves do between barriers: 0 nothing, 1 LDS reads, 2 LDS reads + MFMAs (compile-time: the effect is
                    // sensitive to the partner waves' timing -- with this as a run-time branch the probe found 0 of 2.6e9)
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Pre { float4 a, b, c, d; };

template <int MODE>
__device__ __forceinline__ void affine2(float4& v, const floatx2 sc, const floatx2 sh) {   // channel = the pair's HIGH (modes 0, 1) or LOW (2, 3) element
  floatx2 lo = {v.x, v.y}, hi = {v.z, v.w};
  if (MODE == 0) {
    asm volatile("v_pk_fma_f32 %0, %0, %2, %3 op_sel:[0,1,1]\n\tv_pk_fma_f32 %1, %1, %2, %3 op_sel:[0,1,1]" : "+v"(lo), "+v"(hi) : "v"(sc), "v"(sh));
  } else if (MODE == 2) {
    asm volatile("v_pk_fma_f32 %0, %0, %2, %3 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %1, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(lo), "+v"(hi) : "v"(sc), "v"(sh));
  } else {
    const float s = MODE == 1 ? sc[1] : sc[0], h = MODE == 1 ? sh[1] : sh[0];
    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]) : "v"(s), "v"(h));
  }
  v = make_float4(lo[0], lo[1], hi[0], hi[1]);
}

template <int MODE>
__global__ __launch_bounds__(768) void probe(const float4* __restrict__ X, const float* __restrict__ table, float4* __restrict__ Y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* As = reinterpret_cast<float*>(lds);                 // [2][256] (scale, shift) tables, as the kernels keep them
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rank = blockIdx.x, G = gridDim.x;
  const int nloc = (ntiles - rank + G - 1) / G;
  if (wave >= 4) {   // the multipliers' side: LDS fragment reads + MFMAs, one barrier per tile
    const char* base = lds + 4096 + lane * 16;
    floatx4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    for (int i = 0; i < nloc; ++i) {
#if PARTNER
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uintx4 v = *reinterpret_cast<const volatile uintx4*>(base + ((k * 1024 + i * 8192) & 65535));
#if PARTNER == 2
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
#else
        acc[k & 3][0] += __uint_as_float(v[0]);
#endif
      }
#endif
      __syncthreads();
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.678f) Y[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  // stagers: tile t = 2048 float4 (32 KB): thread tid takes float4 tid + 256 k, k = 0 .. 7 (two sets of four rows as in the kernels)
  const int cg = tid >> 4;                                   // 0 .. 15: channel group of the first half, + 16 of the second
  int qi = 0;
  auto issue = [&](Pre& p, Pre& q) {
    const int t = rank + (qi < nloc ? qi : nloc - 1) * G;
    const float4* Xt = X + (size_t)t * 2048 + tid;
    p.a = Xt[0]; p.b = Xt[256]; p.c = Xt[512]; p.d = Xt[768];
    q.a = Xt[1024]; q.b = Xt[1280]; q.c = Xt[1536]; q.d = Xt[1792];
    ++qi;
  };
  int li = 0;
  auto land = [&](Pre p, Pre q) {
    const int t = rank + (li < nloc ? li : nloc - 1) * G;
    const float* Ab = As + (li & 1) * 256 + 4 * cg;
    const float4 sc = *reinterpret_cast<const float4*>(Ab), sh = *reinterpret_cast<const float4*>(Ab + 128);
    const float4 sc2 = *reinterpret_cast<const float4*>(Ab + 64), sh2 = *reinterpret_cast<const float4*>(Ab + 192);
    // rows a, c use the pair (x, y); rows b, d the pair (z, w): every row is one channel
    affine2<MODE>(p.a, floatx2{sc.x, sc.y}, floatx2{sh.x, sh.y}); affine2<MODE>(p.b, floatx2{sc.z, sc.w}, floatx2{sh.z, sh.w});
    affine2<MODE>(p.c, floatx2{sc.x, sc.y}, floatx2{sh.x, sh.y}); affine2<MODE>(p.d, floatx2{sc.z, sc.w}, floatx2{sh.z, sh.w});
    affine2<MODE>(q.a, floatx2{sc2.x, sc2.y}, floatx2{sh2.x, sh2.y}); affine2<MODE>(q.b, floatx2{sc2.z, sc2.w}, floatx2{sh2.z, sh2.w});
    affine2<MODE>(q.c, floatx2{sc2.x, sc2.y}, floatx2{sh2.x, sh2.y}); affine2<MODE>(q.d, floatx2{sc2.z, sc2.w}, floatx2{sh2.z, sh2.w});
    {   // (unconditional straight-line code: a load or store behind a loop-dependent branch makes hipcc drain the queue with vmcnt(0)
        // in front of the packed instructions, and then nothing is in flight while they execute)
      float4* Yt = Y + (size_t)t * 2048 + tid;
      Yt[0] = p.a; Yt[256] = p.b; Yt[512] = p.c; Yt[768] = p.d; Yt[1024] = q.a; Yt[1280] = q.b; Yt[1536] = q.c; Yt[1792] = q.d;
    }
    // the next tile's table (one float per thread), written for the landing after the barrier
    As[((li + 1) & 1) * 256 + tid] = table[(size_t)((t + G) % 64) * 256 + tid];
    ++li;
  };
  As[tid] = table[(size_t)(rank % 64) * 256 + tid];
  __syncthreads();
  Pre p0, q0, p1, q1, p2, q2;
  issue(p0, q0); issue(p1, q1); issue(p2, q2);
  for (int i = 0; i < nloc; i += 3) {                        // nloc % 3 == 0 (host)
    land(p0, q0); issue(p0, q0); __syncthreads();
    land(p1, q1); issue(p1, q1); __syncthreads();
    land(p2, q2); issue(p2, q2); __syncthreads();
  }
}

template <int MODE> static void run(const float4* X, const float* T, float4* Y, int ntiles) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4096 + 65536);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(768), 4096 + 65536, 0, X, T, Y, ntiles);
}

int main(int argc, char** argv) {
  const int repeats = argc > 1 ? atoi(argv[1]) : 20;
  const int ntiles = 256 * 63;                               // x 32 KB = 504 MB; 63 tiles per workgroup
  const size_t nx = (size_t)ntiles * 2048;
  std::vector<float4> h(nx);
  unsigned s = 777;
  for (size_t i = 0; i < nx; ++i) { float v[4]; for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; v[k] = (float)((int)(s >> 8) % 20001 - 10000) * 1e-3f; } h[i] = make_float4(v[0], v[1], v[2], v[3]); }
  std::vector<float> ht(64 * 256);
  for (size_t i = 0; i < ht.size(); ++i) { s = s * 1664525u + 1013904223u; ht[i] = 0.5f + (float)(s >> 12) * (1.f / 1048576.f); }
  float4 *X, *Y0, *Y1; float* T;
  (void)hipMalloc(&X, nx * sizeof(float4)); (void)hipMalloc(&Y0, nx * sizeof(float4)); (void)hipMalloc(&Y1, nx * sizeof(float4)); (void)hipMalloc(&T, ht.size() * sizeof(float));
  (void)hipMemcpy(X, h.data(), nx * sizeof(float4), hipMemcpyHostToDevice); (void)hipMemcpy(T, ht.data(), ht.size() * sizeof(float), hipMemcpyHostToDevice);
  std::vector<float4> a(nx), b(nx);
  unsigned long long bad_hi = 0, bad_lo = 0, bad_scalar = 0;
  int shown = 0;
  for (int r = 0; r < repeats; ++r) {
    run<1>(X, T, Y1, ntiles); (void)hipDeviceSynchronize(); (void)hipMemcpy(b.data(), Y1, nx * sizeof(float4), hipMemcpyDeviceToHost);
    run<0>(X, T, Y0, ntiles); (void)hipDeviceSynchronize(); (void)hipMemcpy(a.data(), Y0, nx * sizeof(float4), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < nx; ++i)
      if (memcmp(&a[i], &b[i], 16) != 0) {
        ++bad_hi;
        // which registers did the instruction compute with?  float4 i = tile t, slot k (row), thread tid; its channel pair of the table
        const size_t t = i / 2048; const int k = (int)((i % 2048) / 256), tid = (int)(i % 256), cg = tid >> 4;
        const int g = (int)(t % 256), li = (int)(t / 256);                 // workgroup, its li-th tile
        const int trow = li == 0 ? g % 64 : (int)((g + (size_t)(li - 1) * 256 + 256) % 64);   // the table the tile saw (see the kernel)
        const int ch = 4 * cg + (k >= 4 ? 64 : 0) + ((k & 1) ? 2 : 0);      // the pair's LOW channel; the form uses ch + 1
        const float* T = &ht[(size_t)trow * 256];
        const float scl = T[ch], sch = T[ch + 1], shl = T[128 + ch], shh = T[128 + ch + 1];
        const float x[4] = {h[i].x, h[i].y, h[i].z, h[i].w}, got[4] = {a[i].x, a[i].y, a[i].z, a[i].w}, want[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
        if (shown < 3) {
          ++shown;
          printf("  repeat %d tile %zu row %d thread %3d:", r, t, k, tid);
          for (int e = 0; e < 4; ++e) {
            const char* how = got[e] == want[e] ? "ok" : got[e] == fmaf(x[e], scl, shl) ? "LOW scale, LOW shift" : got[e] == fmaf(x[e], scl, shh) ? "LOW scale" :
                              got[e] == fmaf(x[e], sch, shl) ? "LOW shift" : "other";
            printf(" [%d] %s", e, how);
          }
          printf("\n      x = %.6g %.6g  got %.9g want %.9g | scale lo %.6g hi %.6g shift lo %.6g hi %.6g | (got - shift_hi) / x = %.6g, (got - shift_lo) / x = %.6g, (got - want) = %.6g\n",
                 x[0], x[1], got[0], want[0], scl, sch, shl, shh, (got[0] - shh) / x[0], (got[0] - shl) / x[0], got[0] - want[0]);
        }
      }
    run<1>(X, T, Y0, ntiles); (void)hipDeviceSynchronize(); (void)hipMemcpy(a.data(), Y0, nx * sizeof(float4), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < nx; ++i) bad_scalar += memcmp(&a[i], &b[i], 16) != 0;
    run<3>(X, T, Y1, ntiles); (void)hipDeviceSynchronize(); (void)hipMemcpy(b.data(), Y1, nx * sizeof(float4), hipMemcpyDeviceToHost);
    run<2>(X, T, Y0, ntiles); (void)hipDeviceSynchronize(); (void)hipMemcpy(a.data(), Y0, nx * sizeof(float4), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < nx; ++i) bad_lo += memcmp(&a[i], &b[i], 16) != 0;
  }
  const unsigned long long tot = (unsigned long long)repeats * nx;
  printf("float4 results that differ from the scalar form's, of %llu each:\n  v_pk_fma_f32 op_sel:[0,1,1] (high register broadcast): %llu\n"
         "  v_pk_fma_f32 op_sel_hi:[1,0,0] (low register broadcast):  %llu\n  scalar form against itself:                                %llu\n", tot, bad_hi, bad_lo, bad_scalar);
  return 0;
}
