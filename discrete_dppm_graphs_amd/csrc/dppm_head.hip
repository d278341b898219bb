// Fused read-out head of the H = 128 PPGN (see dppm_head.h).
#include "dppm_head.h"
#include "dppm_wsplit.h"
#include <cstdlib>

namespace dppm {

// weight matrices -> per-lane MFMA fragments of bf16 pieces, once per forward call (576 KB for L = 8)
template <int NS>
__global__ __launch_bounds__(512) void head_wimg_kernel(const float* __restrict__ cat_w, int ldw, const float* __restrict__ after_w,
                                                                int L, uint4* __restrict__ wimg) {
  const int src = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w + ci;
  uintx4 wh[4], wl[4];
  if (src < L) w3_load_w<NS, false>(wh, wl, cat_w + 2 + (size_t)src * WS_H, ldw, ch, g);
  else w3_load_w<NS, false>(wh, wl, after_w, WS_H, ch, g);
  uint4* o = wimg + (size_t)src * HD_WFRAG + (size_t)w * 2 * 4 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    o[s * 64] = make_uint4(wh[s][0], wh[s][1], wh[s][2], wh[s][3]);
    o[(4 + s) * 64] = make_uint4(wl[s][0], wl[s][1], wl[s][2], wl[s][3]);
  }
}

// one X tile in flight per staging thread (named members: stays in registers): channel groups cg (a .. d) and cg + 16 (e .. h);
// t: float `tid` of the NEXT tile's 256-float (scale, shift) table (NRM); s: the epilogue's per-position input of staging wave k at
// position p0 + lane -- k = 0: A, 1: sigma I, 2: the mask (3: unused) -- written to LDS when the tile belongs to source 0
struct HdPre { float4 a, b, c, d, e, f, g, h; float t, s; };
struct HdW { uintx4 h[4], l[4]; };              // a wave's fragments of one weight phase

template <int NS>
__device__ __forceinline__ void hd_load_w(HdW& W, const uint4* __restrict__ p) {   // p: wimg + phase * HD_WFRAG + wave * 512 + lane
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 q = p[s * 64];
    W.h[s][0] = q.x; W.h[s][1] = q.y; W.h[s][2] = q.z; W.h[s][3] = q.w;
    if (NS > 1) {
      const uint4 r = p[(4 + s) * 64];
      W.l[s][0] = r.x; W.l[s][1] = r.y; W.l[s][2] = r.z; W.l[s][3] = r.w;
    } else {
      W.l[s] = W.h[s];
    }
  }
}

template <int NS>
__device__ __forceinline__ void hd_load_w1(HdW& W, const uint4* __restrict__ p, int s) {   // k step s only
  const uint4 q = p[s * 64];
  W.h[s][0] = q.x; W.h[s][1] = q.y; W.h[s][2] = q.z; W.h[s][3] = q.w;
  if (NS > 1) {
    const uint4 r = p[(4 + s) * 64];
    W.l[s][0] = r.x; W.l[s][1] = r.y; W.l[s][2] = r.z; W.l[s][3] = r.w;
  } else {
    W.l[s] = W.h[s];
  }
}

// w3_mma4 for the LAST tile of a weight phase: as soon as k step s has been issued for all four position groups its fragment
// registers are dead, and the next phase's fragments of that k step are requested into them (no second register set; the L2
// latency hides behind the remaining k steps and the landing of the next tile)
template <int NS>
__device__ __forceinline__ void hd_mma4_reload(floatx4 (&acc)[4], HdW& W, const char* __restrict__ T, const W3Addr& ad, const uint4* __restrict__ next) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = w3_off(ad, j, s);
      const uintx4 xh = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) {
        const uintx4 xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
        acc[j] = w3_mfma(W.h[s], xl, acc[j]);
        acc[j] = w3_mfma(W.l[s], xh, acc[j]);
      }
      acc[j] = w3_mfma(W.h[s], xh, acc[j]);
    }
    hd_load_w1<NS>(W, next, s);
  }
}

// Roles: waves 0 .. 3 stage (global -> registers -> bf16 images in LDS; three tiles in flight per thread, two channel groups each,
// no accumulators), waves 4 .. 11 multiply (weight fragments + accumulators, LDS reads and MFMAs only) and run the epilogue:
// 12 waves at 168 registers (round 2's 8 + 8 waves at 128 registers spilled 36 / 39 registers in the multiplying waves).  Every wave
// executes the same sequence of barriers: one per X tile, HD_NT + 1 per epilogue.
// NRM: the layer outputs in CAT are the RAW m4 results; their masked instance norm (a.aff: [L][B][2][128] scale, shift) is applied
// while staging, and the mask (a.maskp) when the accumulators are finished (a masked u_l is an exact zero: model/ppgn.py:315).
// The (scale, shift) table of a tile's (source, graph) goes through LDS as in w4_mlp_kernel: with tile q the stagers request one
// float4 of tile q+1's table (pre.t), write it to As[(q+1) & 1] when tile q lands and read their channels' pairs one barrier later.
// The epilogue's per-position inputs (A, sigma I, mask) travel the same way: requested with every tile, written to Ss by the
// first staging wave when a tile of source 0 lands, read by the multiplying waves in that super-tile's epilogue.
template <int NS, bool NRM>
__global__ __launch_bounds__(HD_THREADS) void head3_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_hd[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  char* Xs = smem_hd;                                        // [2][BUF] source tile images
  char* Hs = smem_hd + 2 * BUF;                              // [2][BUF] R tile images (epilogue)
  float* Cs = reinterpret_cast<float*>(smem_hd + 4 * BUF);   // [5][128]: cat_w[:,0], cat_w[:,1], cat_b, after_b, final_w
  float* Ls = Cs + 5 * WS_H;                                 // [2][8 waves][64] partial logits
  float* Ss = Ls + 2 * 8 * WS_TP;                            // [2 super-tile parities][HD_NT][3][64]: A, sigma I, mask rows of the super-tile's tiles
  float* As = Ss + 2 * HD_NT * 3 * WS_TP;                    // [2 tile parities][256]: (scale, shift) tables (NRM)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave - HD_STG;                               // < 0: stager, else multiplier of channels 16 w ..
  const int g = lane >> 4, ci = lane & 15;
  const int P = a.P, L = a.L;
  const int nsp = (P + HD_SP - 1) / HD_SP;                   // super-tiles per graph
  const int total = a.B * nsp;
  const int rank = blockIdx.x, G = gridDim.x;
  if (rank >= total) return;                                 // whole workgroup, before any barrier
  const int nloc = (total - rank + G - 1) / G;               // super-tiles of this workgroup: rank + i * G

  if (tid < WS_H) {
    Cs[tid] = a.cat_w[(size_t)tid * a.ldw];
    Cs[WS_H + tid] = a.cat_w[(size_t)tid * a.ldw + 1];
    Cs[2 * WS_H + tid] = a.cat_b[tid];
    Cs[3 * WS_H + tid] = a.after_b[tid];
    Cs[4 * WS_H + tid] = a.final_w[tid];
  }
  if (NRM && tid < 64)                                       // tile 0's table, directly (source 0, the graph of super-tile `rank`)
    *reinterpret_cast<float4*>(As + 4 * tid) = *reinterpret_cast<const float4*>(a.aff + (size_t)(rank / nsp) * 256 + 4 * tid);
  const float fbias = a.final_b[0];
  const int per_sp = L * HD_NT;                              // X tiles per super-tile
  const int nq = nloc * per_sp;                              // X tiles of this workgroup, in stream order q = (i * L + s) * HD_NT + t

  auto flush = [&](int b, int sp0, int t) {                  // logits of tile t from the 8 multiplier waves' partial sums (threads 0 .. 63)
    const float* Lb = Ls + (t & 1) * 8 * WS_TP + tid;
    float v = fbias;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += Lb[k * WS_TP];
    const int p = sp0 + WS_TP * t + tid;
    const int row = p / a.ld, col = p - row * a.ld;
    if (p < P && col < a.N) a.logits[((size_t)b * a.N + row) * a.N + col] = v;
  };

  if (w < 0) {
    // ------------------------------------------------------------------------------------------------ stagers
    const unsigned sq = w4_q(tid), scg = w4_cg(tid);         // position quad 0 .. 15, channel groups scg and scg + 16
    const unsigned o0 = (4 * scg) * P + 4 * sq;
    const int oc = 4 * sq;
    int qi = 0, qs = 0, qt = 0;                              // issue cursor
    auto issue = [&](HdPre& pre) {                           // unconditional clamped loads of the tile at the cursor
      const int ii = qi < nloc ? qi : nloc - 1;
      const int st = rank + ii * G;
      const int b = __builtin_amdgcn_readfirstlane(st / nsp);
      int p0 = __builtin_amdgcn_readfirstlane((st - b * nsp) * HD_SP + WS_TP * qt);
      if (p0 >= P) p0 = 0;                                   // tile beyond the plane (ragged last super-tile): harmless re-read
      const float* Xt = a.cat + (size_t)b * a.cat_bs + (size_t)(2 + qs * WS_H) * P + p0 + ((p0 + oc < P) ? o0 : 0u);
      pre.a = *reinterpret_cast<const float4*>(Xt);
      pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
      pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
      pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
      const float* Xu = Xt + (size_t)64 * P;
      pre.e = *reinterpret_cast<const float4*>(Xu);
      pre.f = *reinterpret_cast<const float4*>(Xu + (size_t)P);
      pre.g = *reinterpret_cast<const float4*>(Xu + (size_t)2 * P);
      pre.h = *reinterpret_cast<const float4*>(Xu + (size_t)3 * P);
      {   // the epilogue's side inputs travel with the stream (every tile, unconditionally: the counted waits stay exact)
        const int ps = p0 + lane < P ? p0 + lane : 0;
        const float* Sd = (NRM && wave == 2) ? a.maskp + (size_t)b * P : a.cat + (size_t)b * a.cat_bs + (wave == 1 ? (size_t)P : (size_t)0);   // (uniform)
        pre.s = Sd[ps];
      }
      if (++qt == HD_NT) { qt = 0; if (++qs == L) { qs = 0; ++qi; } }
      if (NRM) {                                             // the cursor now names the NEXT tile: its (source, graph) table
        const int b1 = __builtin_amdgcn_readfirstlane((rank + (qi < nloc ? qi : nloc - 1) * G) / nsp);
        pre.t = a.aff[((size_t)qs * a.B + b1) * 256 + tid];
      }
    };
    int li = 0, ls = 0, lt = 0, lq = 0;                      // landing cursor, and the tile's number in the stream
    auto land = [&](const HdPre& pre, int buf) {
      // No zero fill: positions past the plane's end and landings past the last tile hold clamped duplicates.  A position is a column
      // of D = W X -- no lane of another position reads it -- and flush() stores in-range positions only.
      char* Xb = Xs + buf * BUF;
      const float* Ab = As + (lq & 1) * 256 + 4 * scg;
      {
        float4 xa = pre.a, xb = pre.b, xc = pre.c, xd = pre.d;
        if (NRM) {
          const float4 sc = *reinterpret_cast<const float4*>(Ab), sh = *reinterpret_cast<const float4*>(Ab + 128);
          w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
        }
        w3_tile_store4_qc<NS>(xa, xb, xc, xd, Xb, sq, scg);
      }
      {
        float4 xa = pre.e, xb = pre.f, xc = pre.g, xd = pre.h;
        if (NRM) {
          const float4 sc = *reinterpret_cast<const float4*>(Ab + 64), sh = *reinterpret_cast<const float4*>(Ab + 192);
          w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
        }
        w3_tile_store4_qc<NS>(xa, xb, xc, xd, Xb, sq, scg + 16);
      }
      if (NRM) As[((lq + 1) & 1) * 256 + tid] = pre.t;
      if (ls == 0 && wave < 3) Ss[((li & 1) * HD_NT + lt) * 3 * WS_TP + tid] = pre.s;   // (uniform) rows A, sigma I, mask (NRM; else unused)
      ++lq;
      if (++lt == HD_NT) { lt = 0; if (++ls == L) { ls = 0; ++li; } }
    };
    HdPre p0s, p1s, p2s;                                     // tile k travels in set k % 3, lands in X buffer k & 1
    if (NRM) __syncthreads();                                // tile 0's table is in As[0]
    issue(p0s); issue(p1s); issue(p2s);
    land(p0s, 0);
    __syncthreads();                                         // tile 0 is in X buffer 0
    int left = per_sp;                                       // X tiles until the next epilogue
    for (int q = 0; q < nq; q += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (q + u < nq) {
          // tile q+u is being multiplied: its set (u % 3) is free for tile q+u+3; tile q+u+1 lands in buffer (u+1) & 1
          if (u % 3 == 0) { issue(p0s); land(p1s, (u + 1) & 1); }
          else if (u % 3 == 1) { issue(p1s); land(p2s, (u + 1) & 1); }
          else { issue(p2s); land(p0s, (u + 1) & 1); }
          __syncthreads();
          if (--left == 0) {                                 // end of a super-tile: the multipliers' epilogue barriers, and the logits
            left = per_sp;
            const int i = (q + u) / per_sp;
            const int st = rank + i * G;
            const int b = __builtin_amdgcn_readfirstlane(st / nsp);
            const int sp0 = __builtin_amdgcn_readfirstlane((st - b * nsp) * HD_SP);
#pragma unroll
            for (int t = 0; t < HD_NT; ++t) {
              __syncthreads();
              if (t > 0 && tid < WS_TP) flush(b, sp0, t - 1);
            }
            __syncthreads();
            if (tid < WS_TP) flush(b, sp0, HD_NT - 1);
          }
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- multipliers
  const W3Addr ad = w3_addr(ci, g);
  const unsigned hbase = ci * 256 + (g & 1) * 8, hchunk = (2 * w + (g >> 1)) ^ ci;   // R slot in the hidden image (dppm_wsplit.h)
  const uint4* wbase = a.wimg + (size_t)w * 2 * 4 * 64 + lane;
  floatx4 acc[HD_NT][4];
  HdW W;
  hd_load_w<NS>(W, wbase);                                   // phase 0 of the first super-tile
  if (NRM) __syncthreads();                                  // (the stagers' table of tile 0)
  __syncthreads();                                           // tile 0 is in X buffer 0
  for (int i = 0; i < nloc; ++i) {
#pragma unroll
    for (int t = 0; t < HD_NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int ph = 0; ph < L; ++ph) {                         // source ph: HD_NT tiles; tile t of the phase sits in X buffer t & 1 (HD_NT even)
      const uint4* wnext = wbase + (size_t)(ph + 1) * HD_WFRAG;
#pragma unroll
      for (int t = 0; t < HD_NT; ++t) {
        if (t < HD_NT - 1) w3_mma4<NS, true>(acc[t], W.h, W.l, Xs + (t & 1) * BUF, ad);
        else hd_mma4_reload<NS>(acc[t], W, Xs + (t & 1) * BUF, ad, wnext);   // next phase's fragments arrive behind the last tile
        __syncthreads();
      }
    }
    // ---- epilogue of super-tile i: W holds after_conv
    const int st = rank + i * G;
    const int b = __builtin_amdgcn_readfirstlane(st / nsp);
    const int sp0 = __builtin_amdgcn_readfirstlane((st - b * nsp) * HD_SP);
    const int c4 = 16 * w + 4 * g;                           // the lane's channels c4 .. c4 + 3
#pragma unroll
    for (int t = 0; t < HD_NT; ++t) {
      char* Hb = Hs + (t & 1) * BUF;
      {
        const float4 ws0 = *reinterpret_cast<const float4*>(Cs + c4), ws1 = *reinterpret_cast<const float4*>(Cs + WS_H + c4);
        const float4 cb = *reinterpret_cast<const float4*>(Cs + 2 * WS_H + c4);
        // The side values (and NRM: the mask) of the tile come from LDS, where the stagers left them with source 0 of this super-tile
        // (Ss, by super-tile parity: the next super-tile's rows land before this epilogue).  Round 2 loaded them here from global memory
        // next to reloads of spilled registers and needed an explicit vmcnt(0) to be deterministic; the multiplying waves now issue no
        // vector-memory load in the epilogue besides the weight fragments of the next phase.
        const float* Sr = Ss + ((i & 1) * HD_NT + t) * 3 * WS_TP + ci;
        float av4[4], sv4[4], mk4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { av4[j] = Sr[16 * j]; sv4[j] = Sr[WS_TP + 16 * j]; mk4[j] = NRM ? Sr[2 * WS_TP + 16 * j] : 1.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                        // R = acc + side input + bias -> registers and hidden image
          const float av = av4[j], sv = sv4[j];
          floatx4& r = acc[t][j];
          if (NRM) {
            if (!(mk4[j] > 0.f)) r = floatx4{0.f, 0.f, 0.f, 0.f};
          }
          // (w3_side_s: the compiler's packed form of this term, with crossed operand halves, is what made the kernel nondeterministic)
          r[0] += w3_side_s(ws0.x, ws1.x, cb.x, av, sv); r[1] += w3_side_s(ws0.y, ws1.y, cb.y, av, sv);
          r[2] += w3_side_s(ws0.z, ws1.z, cb.z, av, sv); r[3] += w3_side_s(ws0.w, ws1.w, cb.w, av, sv);
          unsigned h01, l01 = 0, h23, l23 = 0;
          w3_split2<NS>(r[0], r[1], h01, l01);
          w3_split2<NS>(r[2], r[3], h23, l23);
          const unsigned off = hbase + j * 4096 + ((hchunk ^ j) << 4);
          *reinterpret_cast<uint2*>(Hb + off) = make_uint2(h01, h23);
          if (NS > 1) *reinterpret_cast<uint2*>(Hb + W3_IMG + off) = make_uint2(l01, l23);
        }
      }
      __syncthreads();
      floatx4 act[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) act[j] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (t < HD_NT - 1) w3_mma4<NS, true>(act, W.h, W.l, Hb, ad);
      else hd_mma4_reload<NS>(act, W, Hb, ad, wbase);        // source 0 of the next super-tile
      const float4 ab = *reinterpret_cast<const float4*>(Cs + 3 * WS_H + c4), fw = *reinterpret_cast<const float4*>(Cs + 4 * WS_H + c4);
      float part[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx4& r = acc[t][j];
        float v = fw.x * (r[0] + ws_leaky(act[j][0] + ab.x)) + fw.y * (r[1] + ws_leaky(act[j][1] + ab.y)) +
                  fw.z * (r[2] + ws_leaky(act[j][2] + ab.z)) + fw.w * (r[3] + ws_leaky(act[j][3] + ab.w));
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        part[j] = v;
      }
      const float mine = g == 0 ? part[0] : (g == 1 ? part[1] : (g == 2 ? part[2] : part[3]));
      Ls[(t & 1) * 8 * WS_TP + w * WS_TP + 16 * g + ci] = mine;
    }
    __syncthreads();
  }
}

template <int NS> inline size_t head3_lds() { return (size_t)4 * (NS > 1 ? 2 : 1) * W3_IMG + (5 * WS_H + 2 * 8 * WS_TP + 2 * HD_NT * 3 * WS_TP + 2 * 256) * sizeof(float); }

template <int NS>
static int head3_launch(const HeadArgs& a, const float* after_w, hipStream_t st) {
  const size_t lds = head3_lds<NS>();
  const bool nrm = a.aff != nullptr;
  if (nrm && !a.maskp) return fail(DPPM_E_INVALID, "head3: the fused instance norm needs the mask planes");
  auto kern = nrm ? head3_kernel<NS, true> : head3_kernel<NS, false>;
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), lds, "head3"));
  if ((long)(2 + a.L * WS_H) * a.P >= (1L << 31)) return fail(DPPM_E_INVALID, "head3: P=%d too large", a.P);
  hipLaunchKernelGGL(head_wimg_kernel<NS>, dim3(a.L + 1), dim3(512), 0, st, a.cat_w, a.ldw, after_w, a.L, const_cast<uint4*>(a.wimg));
  DPPM_TRY(check_launch("head_wimg"));
  const long total = (long)a.B * ((a.P + HD_SP - 1) / HD_SP);
  const int grid = (int)(total < 256 ? total : 256);     // one workgroup per CU
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(HD_THREADS), lds, st, a);
  if (prof) prof_after(st, PROF_HEAD, prof_arith_of_pieces(NS), 2.0 * WS_H * ((double)a.L * WS_H + 2 + WS_H + 1) * a.P * a.B,
                       sizeof(float) * ((double)a.L * WS_H + 2) * a.P * a.B + sizeof(float) * (double)a.N * a.N * a.B);
  return check_launch("head3");
}

int head3_run(const HeadArgs& a, const float* after_w, int ns, hipStream_t st) {
  return ns == 3 ? head3_launch<3>(a, after_w, st) : head3_launch<1>(a, after_w, st);
}

}  // namespace dppm
