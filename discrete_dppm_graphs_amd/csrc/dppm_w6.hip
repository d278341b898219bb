// Float32-grade H = 128 linears for TRAINING on the bf16 matrix pipe (see dppm_w6.h): Y (+)= epilogue(sum_s W_s X_s + b) with every
// operand split into three bf16 pieces (x = hi + mid + lo: all 24 significand bits) and the six products down to 2^-16 per pair
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; smallest first, float32 accumulation) -- the split of the weight gradients
// (dppm_wgrad.h) and of the training contraction (dppm_plane4.hip).  Six bf16 MFMAs take 6/16 of the time of the exact-f32 MFMAs
// they replace, which moves these kernels (61-75 % of the f32 matrix peak in dppm_wstat.h) to the HBM roof.
//
// Organisation as in dppm_w4.hip: waves 0-7 stage (three tiles in flight per thread; the split happens when a tile lands in its
// three LDS images), waves 8-15 multiply: 16 output channels per wave, the matrices' three pieces in 48 registers per source,
// X fragments from LDS, 16-byte stores.  One barrier per staged tile.  Stream order q = NSRC i + s (tile i, source s).  Up to
// WS_MAX_JOBS jobs per launch (a workgroup serves one job; jobs that read the same source sit on one XCD).
#include "dppm_w6.h"
#include "dppm_wsplit.h"

namespace dppm {

constexpr int W6L_THREADS = 1024;
constexpr int W6L_BUF = 3 * W3_IMG;                           // bytes of the three piece images of one tile (48 KB)

struct W6Pre { float4 a, b, c, d; };                         // one tile in flight per staging thread (named members: registers)

// (x0, x1) -> three packed bf16 pairs
__device__ __forceinline__ void w6_split(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = w3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = w3_pack(r0, r1);
  l = w3_pack(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void w6_put4(char* __restrict__ T, unsigned off, float x0, float x1, float x2, float x3) {
  unsigned h01, m01, l01, h23, m23, l23;
  w6_split(x0, x1, h01, m01, l01);
  w6_split(x2, x3, h23, m23, l23);
  *reinterpret_cast<uint2*>(T + off) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(T + W3_IMG + off) = make_uint2(m01, m23);
  *reinterpret_cast<uint2*>(T + 2 * W3_IMG + off) = make_uint2(l01, l23);
}
// w3_tile_store4 with three pieces (same image layout and swizzle)
__device__ __forceinline__ void w6_tile_store4(float4 r0, float4 r1, float4 r2, float4 r3, char* __restrict__ T, int tid) {
  const unsigned q = tid & 15, cg = tid >> 4;
  const unsigned base = (4 * q) * 256 + (cg & 1) * 8;
  const unsigned kc = (cg >> 1) ^ (4 * (q & 3)) ^ (q >> 2);
  w6_put4(T, base + ((kc ^ 0) << 4), r0.x, r1.x, r2.x, r3.x);
  w6_put4(T, base + 256 + ((kc ^ 1) << 4), r0.y, r1.y, r2.y, r3.y);
  w6_put4(T, base + 512 + ((kc ^ 2) << 4), r0.z, r1.z, r2.z, r3.z);
  w6_put4(T, base + 768 + ((kc ^ 3) << 4), r0.w, r1.w, r2.w, r3.w);
}
// a lane's weights of output channel ch as four K = 32 fragments per piece (w3_load_w with three pieces)
template <bool TRANS>
__device__ __forceinline__ void w6_load_w(uintx4 (&wh)[4], uintx4 (&wm)[4], uintx4 (&wl)[4], const float* __restrict__ W, int ldw, int ch, int g) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = TRANS ? W[(size_t)(32 * s + 8 * g + e) * ldw + ch] : W[(size_t)ch * ldw + 32 * s + 8 * g + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h, m, l;
      w6_split(v[2 * e], v[2 * e + 1], h, m, l);
      wh[s][e] = h; wm[s][e] = m; wl[s][e] = l;
    }
  }
}
// acc[j] += X-tile(positions 16 j ..) x W-slice over K = 128, D = X^T W^T (lane: 4 positions of one channel).  AHEAD: fragment reads
// one (k step, position group) ahead of their MFMAs (12 more registers: the single-source form); the order is pinned either way.
template <bool AHEAD>
__device__ __forceinline__ void w6_mma4(floatx4 (&acc)[4], const uintx4 (&wh)[4], const uintx4 (&wm)[4], const uintx4 (&wl)[4],
                                        const char* __restrict__ T, const W3Addr& ad) {
  uintx4 xh, xm, xl;
  if (AHEAD) {
    xh = *reinterpret_cast<const uintx4*>(T + w3_off(ad, 0, 0));
    xm = *reinterpret_cast<const uintx4*>(T + W3_IMG + w3_off(ad, 0, 0));
    xl = *reinterpret_cast<const uintx4*>(T + 2 * W3_IMG + w3_off(ad, 0, 0));
  }
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int s = n >> 2, j = n & 3;
    uintx4 ch_, cm_, cl_;
    if (AHEAD) {
      ch_ = xh; cm_ = xm; cl_ = xl;
      if (n + 1 < 16) {
        const unsigned off = w3_off(ad, (n + 1) & 3, (n + 1) >> 2);
        xh = *reinterpret_cast<const uintx4*>(T + off);
        xm = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
        xl = *reinterpret_cast<const uintx4*>(T + 2 * W3_IMG + off);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
    } else {
      const unsigned off = w3_off(ad, j, s);
      ch_ = *reinterpret_cast<const uintx4*>(T + off);
      cm_ = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
      cl_ = *reinterpret_cast<const uintx4*>(T + 2 * W3_IMG + off);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    acc[j] = w3_mfma(cm_, wm[s], acc[j]);                     // smallest terms first
    acc[j] = w3_mfma(cl_, wh[s], acc[j]);
    acc[j] = w3_mfma(ch_, wl[s], acc[j]);
    acc[j] = w3_mfma(cm_, wh[s], acc[j]);
    acc[j] = w3_mfma(ch_, wm[s], acc[j]);
    acc[j] = w3_mfma(ch_, wh[s], acc[j]);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
  }
}

template <int NSRC, bool TRANS>
__global__ __launch_bounds__(W6L_THREADS) void w6_lin_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem6[];   // [2][W6L_BUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w8 + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  if (who.rank >= ntiles) return;                            // whole workgroup, before any barrier
  const int nloc = (ntiles - who.rank + who.per_job - 1) / who.per_job;   // tiles rank + i * per_job
  const int nq = NSRC * nloc;

  if (role == 0) {
    // ------------------------------------------------------------------------------------------------ stagers
    const unsigned o0 = (unsigned)(4 * (tid >> 4)) * P + 4 * (tid & 15);   // channels 4 cg .., positions 4 q ..
    const int oc = 4 * (tid & 15);
    int qi = 0, qs = 0;                                      // issue cursor (tile, source)
    auto issue = [&](W6Pre& pre) {                           // unconditional clamped loads of the tile at the cursor
      const int t = who.rank + (qi < nloc ? qi : nloc - 1) * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
      const float* Xt = ((NSRC == 2 && qs) ? J.X2 + (size_t)b * J.x2_bs : J.X + (size_t)b * J.x_bs) + p0 + ((p0 + oc < P) ? o0 : 0u);
      pre.a = *reinterpret_cast<const float4*>(Xt);
      pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
      pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
      pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
      if (++qs == NSRC) { qs = 0; ++qi; }
    };
    int li = 0, ls = 0;                                      // landing cursor
    auto land = [&](const W6Pre& pre, int buf) {
      const int t = who.rank + (li < nloc ? li : nloc - 1) * who.per_job;
      const int b = t / tpp, p0 = (t - b * tpp) * WS_TP;
      const bool in = li < nloc && p0 + oc < P;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      w6_tile_store4(in ? pre.a : z, in ? pre.b : z, in ? pre.c : z, in ? pre.d : z, smem6 + buf * W6L_BUF, tid);
      if (++ls == NSRC) { ls = 0; ++li; }
    };
    W6Pre p0s, p1s, p2s;
    issue(p0s); issue(p1s); issue(p2s);
    land(p0s, 0);
    __syncthreads();                                         // tile 0 is in buffer 0
    for (int q = 0; q < nq; q += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (q + u < nq) {
          // tile q+u is being multiplied: its set (u % 3) is free for tile q+u+3; tile q+u+1 lands in buffer (u+1) & 1
          if (u % 3 == 0) { issue(p0s); land(p1s, (u + 1) & 1); }
          else if (u % 3 == 1) { issue(p1s); land(p2s, (u + 1) & 1); }
          else { issue(p2s); land(p0s, (u + 1) & 1); }
          __syncthreads();
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- multipliers
  uintx4 wh[NSRC][4], wm[NSRC][4], wl[NSRC][4];
  w6_load_w<TRANS>(wh[0], wm[0], wl[0], J.W1, J.ldw1, ch, g);
  if (NSRC == 2) w6_load_w<TRANS>(wh[NSRC - 1], wm[NSRC - 1], wl[NSRC - 1], J.W2, J.ldw2, ch, g);
  const float bv = J.b1 ? J.b1[ch] : 0.f;
  const W3Addr ad = w3_addr(ci, g);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  __syncthreads();                                           // tile 0 is in buffer 0
  int q = 0;
  for (int i = 0; i < nloc; ++i) {
    const int t = who.rank + i * who.per_job;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSRC; ++s, ++q) {
      w6_mma4<NSRC == 1>(acc, wh[s], wm[s], wl[s], smem6 + (q & 1) * W6L_BUF, ad);
      if (s == NSRC - 1) {
        if (J.Hsaved) {   // backward of an activation: the result times LeakyReLU'(saved activation) (uniform branch)
          const float* Ht = J.Hsaved + (size_t)b * J.hs_bs + p0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if ((int)m0 + 16 * j < P - p0) {
              const float4 h = *reinterpret_cast<const float4*>(Ht + r0 + 16 * j);
              acc[j][0] *= h.x > 0.f ? 1.f : kSlope; acc[j][1] *= h.y > 0.f ? 1.f : kSlope;
              acc[j][2] *= h.z > 0.f ? 1.f : kSlope; acc[j][3] *= h.w > 0.f ? 1.f : kSlope;
            }
          }
        }
        // (256-byte row pieces through ws_store_wide's DPP transposition, saved activation and old value read in that shape too: slower
        // here, 0.64 against 0.61 ms over the MLP backward launches, 12.07 against 12.31 training steps/s)
        ws_store_out(acc, J.Y + (size_t)b * J.y_bs + p0, r0, a.maskp ? a.maskp + (size_t)b * P + p0 : nullptr, m0, P - p0, bv, a.act != 0,
                     J.accumulate != 0);
      }
      __syncthreads();
    }
  }
}

template <int NSRC, bool TRANS>
static int w6_lin_launch(const WsArgs& a, int prof_class, hipStream_t st) {
  const size_t lds = (size_t)2 * W6L_BUF;
  auto kern = w6_lin_kernel<NSRC, TRANS>;
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds), "w6_lin"));
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  const long slots = w3_slots(a.njobs, ntiles);              // one workgroup per CU: 32 per XCD, jobs that share X on one XCD
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.njobs * slots)), dim3(W6L_THREADS), lds, st, a);
  if (prof) prof_after(st, prof_class, AR_F32X6, 2.0 * NSRC * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, NSRC, false, a.job[0].Hsaved != nullptr));
  return check_launch("w6_lin");
}

bool w6_lin_ok(const WsArgs& a, int nsrc) {
  if (a.njobs < 1 || a.njobs > WS_MAX_JOBS || a.P <= 0 || a.P % 4 != 0 || a.B <= 0 || (long)WS_H * a.P >= (1L << 31) || a.side || a.aff || a.stat ||
      a.inl_w || (a.maskp && !ws_al16(a.maskp)) || nsrc != 1)   // two sources: 96 weight registers, 45+ spills at 128: not offered
    return false;
  for (int i = 0; i < a.njobs; ++i) {
    const WsJob& j = a.job[i];
    if (!j.X || !j.W1 || !j.Y || !ws_al16(j.X) || !ws_al16(j.Y) || j.x_bs % 4 || j.y_bs % 4) return false;
    if (j.Hsaved && (!ws_al16(j.Hsaved) || j.hs_bs % 4 || a.act || j.b1)) return false;   // the derivative form has no bias / activation of its own
    if (nsrc == 2 && (!j.X2 || !j.W2 || !ws_al16(j.X2) || j.x2_bs % 4)) return false;
  }
  return true;
}

int w6_lin_run(const WsArgs& a, int nsrc, bool trans, hipStream_t st, int prof_class) {
  if (!w6_lin_ok(a, nsrc)) return fail(DPPM_E_INVALID, "w6_lin: unsupported arguments");
  if (prof_class < 0) prof_class = PROF_WS_COLS;
  return trans ? w6_lin_launch<1, true>(a, prof_class, st) : w6_lin_launch<1, false>(a, prof_class, st);
}

}  // namespace dppm
