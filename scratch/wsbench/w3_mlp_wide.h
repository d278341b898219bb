// Experiment: the split MLP with 32 channels x 32 positions per wave (wave = channel group cg, position half ph) instead of
// 16 channels x 64 positions: every X / H fragment read from LDS feeds two channel tiles, halving the LDS read traffic.
#pragma once
#include "dppm_wsplit.h"
namespace dppm {
// acc[t] += tile(positions of fragment slot jj) x W-slice of channel tile t over K = 128
template <int NS, bool W_IS_A>
__device__ __forceinline__ void w3w_mma(floatx4 (&acc)[2], const uintx4 (&wh)[2][4], const uintx4 (&wl)[2][4], const char* __restrict__ T,
                                        const W3Addr& ad, int jj) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned off = w3_off(ad, jj, s);
    const uintx4 xh = *reinterpret_cast<const uintx4*>(T + off);
    uintx4 xl = xh;
    if (NS > 1) xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (NS > 1) {
        acc[t] = W_IS_A ? w3_mfma(wh[t][s], xl, acc[t]) : w3_mfma(xl, wh[t][s], acc[t]);
        acc[t] = W_IS_A ? w3_mfma(wl[t][s], xh, acc[t]) : w3_mfma(xh, wl[t][s], acc[t]);
      }
      acc[t] = W_IS_A ? w3_mfma(wh[t][s], xh, acc[t]) : w3_mfma(xh, wh[t][s], acc[t]);
    }
  }
}

template <int NS, int DEPTH>
__global__ __launch_bounds__(W3M_THREADS) void w3w_mlp_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  char* Xs = smem3;                       // [2][NIMG] images
  char* Hs = smem3 + 2 * BUF;             // [2][NIMG] images
  float* Ms = reinterpret_cast<float*>(smem3 + 4 * BUF);   // [4][64] mask rows: tile i is written in iteration i-1 and read in i+1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15;
  const int cg = w8 & 3, ph = w8 >> 2;         // this wave: channels 32 cg .. +31, position tiles 2 ph, 2 ph + 1 of the 64-position tile
  const int ch0 = 32 * cg + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  if (who.rank >= ntiles) return;   // whole workgroup, before any barrier

  uintx4 wh[2][4], wl[2][4];
  float bv[2][4];   // stage 1: biases of channels 32 cg + 16 t + 4 g + r; stage 2: bias of channel ch0 + 16 t in bv[t][0]
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    w3_load_w<NS, false>(wh[t], wl[t], role ? J.W2 : J.W1, role ? J.ldw2 : J.ldw1, ch0 + 16 * t, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[t][r] = role ? (J.b2 ? J.b2[ch0 + 16 * t] : 0.f) : (J.b1 ? J.b1[32 * cg + 16 * t + 4 * g + r] : 0.f);
  }

  const unsigned o0 = (unsigned)(4 * ((tid & 511) >> 4)) * P + 4 * (tid & 15);   // staging (waves 0-7): channels 4cg.., positions 4q..
  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch0 * P + 4 * g;
  const unsigned m0 = 4 * g;
  W3Addr ad = w3_addr(ci, g);
  if (ph) { ad.aj[0] = ad.aj[2]; ad.aj[1] = ad.aj[3]; }   // aj[jj] = fragment offset of position tile 2 ph + jj
  // stage 1 result slot in the hidden image: position 16 j + ci, k = 16 w8 + 4 g .. +3 (chunk 2 w8 + (g >> 1), half g & 1)
  const unsigned hbase = ci * 256 + (g & 1) * 8, hchunk = (4 * cg + (g >> 1)) ^ ci;   // channel tile t: chunk ^ (2 t)

  // tiles of this workgroup: t_i = rank + i * per_job, i < nloc (>= 1).
  // Every global load below is UNCONDITIONAL straight-line code (indices clamped instead of predicated): a load inside
  // a lane- or loop-dependent branch makes the compiler drain the whole queue (s_waitcnt vmcnt(0)) at the join, which
  // serialises the prefetch with the compute.
  const int nloc = (ntiles - who.rank + who.per_job - 1) / who.per_job;
  const float* maskp = a.maskp;
  // Only the stage-1 waves stage tiles: their queue then holds loads only, so the counted waits the compiler emits are
  // exact.  (A wave that also stores would have to drain its just-issued stores before every landing.)
  auto issue = [&](int i, W3Pre& pre) {   // global loads of tile min(i, nloc-1) into a prefetch register set
    const int t = who.rank + (i < nloc ? i : nloc - 1) * who.per_job;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const float* Xt = J.X + (size_t)b * J.x_bs + p0 + ((p0 + oc < P) ? o0 : 0u);   // ragged last tile: re-read in-range data, zeroed when landed
    pre.a = *reinterpret_cast<const float4*>(Xt);
    pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
    pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
    pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
    if (maskp) pre.m = maskp[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
  };
  auto land = [&](int i, const W3Pre& pre) {   // register set -> images of buffer i & 1, mask row i & 3
    const int t = who.rank + (i < nloc ? i : nloc - 1) * who.per_job;
    const int p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const bool in = p0 + oc < P;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    w3_tile_store4<NS>(in ? pre.a : z, in ? pre.b : z, in ? pre.c : z, in ? pre.d : z, Xs + (i & 1) * BUF, tid);
    if (tid < 64) Ms[(i & 3) * 64 + tid] = (p0 + lane < P) ? pre.m : 0.f;
  };
  // one iteration: stage 1 of tile i and stage 2 of tile i-1; tile i + AHEAD is requested into (pi, mi) before the MFMAs,
  // tile i+1 is landed from (pl, ml) after them
  auto step = [&](int i, W3Pre& pi, const W3Pre& pl) {
    if (role == 0) {
      issue(i + DEPTH, pi);
      if (i < nloc) {
        char* Hb = Hs + (i & 1) * BUF;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * ph + jj;
          floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
          w3w_mma<NS, true>(acc, wh, wl, Xs + (i & 1) * BUF, ad, jj);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            unsigned h01, l01 = 0, h23, l23 = 0;
            w3_split2<NS>(ws_leaky(acc[t][0] + bv[t][0]), ws_leaky(acc[t][1] + bv[t][1]), h01, l01);
            w3_split2<NS>(ws_leaky(acc[t][2] + bv[t][2]), ws_leaky(acc[t][3] + bv[t][3]), h23, l23);
            const unsigned off = hbase + j * 4096 + ((hchunk ^ (2 * t) ^ j) << 4);
            *reinterpret_cast<uint2*>(Hb + off) = make_uint2(h01, h23);
            if (NS > 1) *reinterpret_cast<uint2*>(Hb + W3_IMG + off) = make_uint2(l01, l23);
          }
        }
      }
      land(i + 1, pl);
    } else if (i >= 1) {
      const int t = who.rank + (i - 1) * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
      const float* Mb = Ms + ((i - 1) & 3) * 64;
      float* Yt = J.Y + (size_t)b * J.y_bs + p0;
      const int lim = P - p0;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * ph + jj;
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        w3w_mma<NS, false>(acc, wh, wl, Hs + ((i - 1) & 1) * BUF, ad, jj);
        const int pj = (int)m0 + 16 * j;
        if (pj < lim) {
          float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
          if (maskp) m = *reinterpret_cast<const float4*>(Mb + pj);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float4 v = ws_f4(acc[t]);
            v.x = (v.x + bv[t][0]) * m.x; v.y = (v.y + bv[t][0]) * m.y; v.z = (v.z + bv[t][0]) * m.z; v.w = (v.w + bv[t][0]) * m.w;
            *reinterpret_cast<float4*>(Yt + r0 + (size_t)(16 * t) * P + 16 * j) = v;
          }
        }
      }
    }
    __syncthreads();
  };
  W3Pre preA, preB;
  preA.m = preB.m = 1.f;
  if (role == 0) {
    issue(0, preA);
    land(0, preA);
    if (DEPTH == 2) issue(1, preB);
  }
  __syncthreads();
  if (DEPTH == 1) {
    for (int i = 0; i <= nloc; ++i) step(i, preA, preA);
  } else {   // tile k travels in set k & 1
    for (int i = 0;;) {
      step(i, preA, preB);
      if (++i > nloc) break;
      step(i, preB, preA);
      if (++i > nloc) break;
    }
  }
}


template <int NS, int DEPTH>
inline int w3w_mlp_forward(const WsArgs& a, hipStream_t st) {
  DPPM_TRY(ws_check(a, "w3w_mlp_forward"));
  static bool set = false;
  const size_t lds = w3m_lds<NS>();
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(w3w_mlp_kernel<NS, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(DPPM_E_LAUNCH, "w3w_mlp_forward: LDS limit");
    set = true;
  }
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  long slots = 32 / a.njobs;
  const long need = (ntiles + 7) / 8;
  if (slots > need) slots = need;
  if (slots < 1) slots = 1;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(w3w_mlp_kernel<NS, DEPTH>), dim3((unsigned)(8 * a.njobs * slots)), dim3(W3M_THREADS), lds, st, a);
  return check_launch("w3w_mlp_forward");
}
}  // namespace dppm
