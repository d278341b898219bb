// Experiment: the split-bf16 fused MLP with the X tile brought in by LDS-DMA (global_load_lds_dwordx4) into a two-slot ring of
// raw float32 tiles that are converted IN PLACE into the hi/lo bf16 images: no prefetch registers, the loads of tile i+2 are
// in flight from the middle of iteration i to the end of iteration i+1.
#pragma once
#include "dppm_wsplit.h"

namespace dppm {

constexpr int W3D_SLOT = 2 * W3_IMG;   // 32 KB: raw tile [128 ch][64 pos] f32, later its hi + lo images

// one wave-instruction: lane l copies 16 bytes from its own global address to LDS byte address lds_dst + 16 l
__device__ __forceinline__ void w3d_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

__global__ __launch_bounds__(W3M_THREADS) void w3d_mlp_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  constexpr int NS = 3;
  char* Ring = smem3;                                  // [2] slots, LDS bytes 0 .. 65535 (the DMA base register holds 16 bits)
  char* Hs = smem3 + 2 * W3D_SLOT;                     // [2][hi, lo] images of the hidden tile
  float* Ms = reinterpret_cast<float*>(smem3 + 4 * W3D_SLOT);   // [4][64] mask rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w8 + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  if (who.rank >= ntiles) return;   // whole workgroup, before any barrier

  uintx4 wh[4], wl[4];
  w3_load_w<NS, false>(wh, wl, role ? J.W2 : J.W1, role ? J.ldw2 : J.ldw1, ch, g);
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = role ? (J.b2 ? J.b2[ch] : 0.f) : (J.b1 ? J.b1[16 * w8 + 4 * g + r] : 0.f);

  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  const W3Addr ad = w3_addr(ci, g);
  const unsigned hbase = ci * 256 + (g & 1) * 8, hchunk = (2 * w8 + (g >> 1)) ^ ci;
  const int nloc = (ntiles - who.rank + who.per_job - 1) / who.per_job;
  const float* maskp = a.maskp;
  const unsigned ring0 = (unsigned)(size_t)Ring;       // LDS byte address of the ring (low 32 bits of the flat address)
  // DMA source of this lane: channel 16 w8 + 4 k + (lane >> 4), positions 4 (lane & 15) .. +3
  const unsigned dsrc = (unsigned)(16 * w8 + (lane >> 4)) * P + 4 * (lane & 15);
  // conversion source of this thread (waves 0-7): channels 4 cg + r, positions 4 q .. +3
  const unsigned craw = (unsigned)(4 * ((tid & 511) >> 4)) * 256 + 16 * (tid & 15);

  float mreg = 1.f;
  auto issue = [&](int i) {   // tile min(i, nloc-1) -> ring slot i & 1 (role 0 only); mask value of position `lane` into mreg (wave 0)
    const int t = who.rank + (i < nloc ? i : nloc - 1) * who.per_job;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const float* Xt = J.X + (size_t)b * J.x_bs + p0 + ((p0 + 4 * (lane & 15) < P) ? dsrc : (unsigned)(16 * w8 + (lane >> 4)) * P);
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + (unsigned)(i & 1) * W3D_SLOT + (unsigned)(16 * w8) * 256);
#pragma unroll
    for (int k = 0; k < 4; ++k) w3d_glds16(Xt + (size_t)(4 * k) * P, dst + (unsigned)k * 1024);
    if (maskp && wave == 0) mreg = maskp[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
  };
  float4 c0, c1, c2, c3;
  auto read_raw = [&](int i) {
    const char* S = Ring + (i & 1) * W3D_SLOT + craw;
    c0 = *reinterpret_cast<const float4*>(S);
    c1 = *reinterpret_cast<const float4*>(S + 256);
    c2 = *reinterpret_cast<const float4*>(S + 512);
    c3 = *reinterpret_cast<const float4*>(S + 768);
  };
  auto write_img = [&](int i) {
    const int t = who.rank + (i < nloc ? i : nloc - 1) * who.per_job;
    const int p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const bool in = p0 + oc < P;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    w3_tile_store4<NS>(in ? c0 : z, in ? c1 : z, in ? c2 : z, in ? c3 : z, Ring + (i & 1) * W3D_SLOT, tid);
  };
  auto put_mask = [&](int i) {   // wave 0: mask row of tile i
    const int t = who.rank + (i < nloc ? i : nloc - 1) * who.per_job;
    const int p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    if (wave == 0) Ms[(i & 3) * 64 + lane] = (p0 + lane < P) ? mreg : 0.f;
  };

  // prologue: tile 0 converted, tile 1 in flight
  if (role == 0) {
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put_mask(0);
  }
  __syncthreads();
  if (role == 0) { read_raw(0); issue(1); }
  __syncthreads();
  if (role == 0) write_img(0);
  __syncthreads();

  for (int i = 0; i <= nloc; ++i) {
    if (role == 0) {
      if (i < nloc) {
        char* Hb = Hs + (i & 1) * W3D_SLOT;
#pragma unroll
        for (int jb = 0; jb < 4; jb += 2) {
          floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
          w3_mma2<NS, true>(acc, wh, wl, Ring + (i & 1) * W3D_SLOT, ad, jb);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unsigned h01, l01 = 0, h23, l23 = 0;
            w3_split2<NS>(ws_leaky(acc[j][0] + bv[0]), ws_leaky(acc[j][1] + bv[1]), h01, l01);
            w3_split2<NS>(ws_leaky(acc[j][2] + bv[2]), ws_leaky(acc[j][3] + bv[3]), h23, l23);
            const unsigned off = hbase + (jb + j) * 4096 + ((hchunk ^ (jb + j)) << 4);
            *reinterpret_cast<uint2*>(Hb + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(Hb + W3_IMG + off) = make_uint2(l01, l23);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile i+1 (and its mask value) has landed
      put_mask(i + 1);
    } else if (i >= 1) {
      const int t = who.rank + (i - 1) * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
      const float* Mb = Ms + ((i - 1) & 3) * 64;
      float* Yt = J.Y + (size_t)b * J.y_bs + p0;
      const int lim = P - p0;
#pragma unroll
      for (int jb = 0; jb < 4; jb += 2) {
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        w3_mma2<NS, false>(acc, wh, wl, Hs + ((i - 1) & 1) * W3D_SLOT, ad, jb);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int pj = (int)m0 + 16 * (jb + j);
          if (pj < lim) {
            float4 v = ws_f4(acc[j]);
            v.x += bv[0]; v.y += bv[0]; v.z += bv[0]; v.w += bv[0];
            if (maskp) {
              const float4 m = *reinterpret_cast<const float4*>(Mb + pj);
              v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            *reinterpret_cast<float4*>(Yt + r0 + 16 * (jb + j)) = v;
          }
        }
      }
    }
    __syncthreads();   // tile i+1 is in LDS from every wave; stage 1 of tile i has left slot i & 1
    if (role == 0) { read_raw(i + 1); issue(i + 2); }
    __syncthreads();   // every raw read done: the slot may be overwritten by the images
    if (role == 0) write_img(i + 1);
    __syncthreads();
  }
  // nothing of this workgroup may still be writing LDS when it ends
  if (role == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline int w3d_mlp_forward(const WsArgs& a, hipStream_t st) {
  DPPM_TRY(ws_check(a, "w3d_mlp_forward"));
  static bool set = false;
  const size_t lds = (size_t)4 * W3D_SLOT + 4 * 64 * sizeof(float);
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(w3d_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(DPPM_E_LAUNCH, "w3d_mlp_forward: cannot raise the dynamic LDS limit to %zu bytes", lds);
    set = true;
  }
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  long slots = 32 / a.njobs;
  const long need = (ntiles + 7) / 8;
  if (slots > need) slots = need;
  if (slots < 1) slots = 1;
  hipLaunchKernelGGL(w3d_mlp_kernel, dim3((unsigned)(8 * a.njobs * slots)), dim3(W3M_THREADS), lds, st, a);
  return check_launch("w3d_mlp_forward");
}

}  // namespace dppm
