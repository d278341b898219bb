// Weight-stationary kernels for the H = 128 PPGN linears (the bulk of the FLOPs: per layer 12 N^2 H^2 of
// 12 N^2 H^2 + 2 H N^3).  In channel-first layout a Linear is Y (128 x P) = W (128 x 128) * X (128 x P) with P = N*ld
// positions per graph: a tall-skinny problem whose only large dimension is P.  So the weights stay put and the
// positions stream through:
//   - persistent workgroups of 8 wave64, TWO per CU (independent barriers: one group's epilogue and staging overlap
//     the other's MFMAs), walk 64-position tiles of all graphs;
//   - wave w keeps the weights of output channels 16w .. 16w+15 in 32 VGPRs per matrix for the whole launch;
//   - the X tile (128 channels x 64 positions, 32 KB) is staged through LDS (global -> registers while the previous
//     tile computes -> LDS) and read by every wave as the MFMA A operand with conflict-free ds_read_b32;
//   - v_mfma_f32_16x16x4_f32 computes D[position][channel] (exact f32), i.e. the transposed tile, so a lane ends up with
//     four CONSECUTIVE positions of one channel: every store is a 16-byte store.  The k order inside the chain is
//     k = 32 * (lane >> 4) + step so that a lane's 32 weights are one contiguous run of its row.
// Kernels
//   ws_mlp_kernel : Linear -> elementwise -> Linear with the intermediate in LDS.
//                   forward  (model/ppgn.py:58-59):  out = (W2 LeakyReLU(W1 X + b1) + b2) * mask         [hid saved for training]
//                   backward:                         dpre = (W1^T dOut) * LeakyReLU'(hid) [saved for the weight gradients],
//                                                     dX (+)= W2^T dpre      (W1 = the MLP's second matrix, W2 = its first)
//   ws_lin_kernel : Y (+)= epilogue(sum_s W_s X_s + b), one or two 128-channel sources (the skip-concat Linear m4 of
//                   model/ppgn.py:66-69 without materialising the cat; after_conv; every transposed data-gradient form).
// A launch carries up to 8 independent jobs (weights / sources / outputs); a workgroup serves one job for its lifetime
// and jobs that read the same source sit on the same XCD so the second read of a tile is an L2 hit.
#pragma once
#include <hip/hip_runtime.h>
#include "dppm_gemm.h"

namespace dppm {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int WS_H = 128;          // channels handled by these kernels
constexpr int WS_TP = 64;          // positions per tile
constexpr int WS_THREADS = 512;
constexpr int WS_XP = 64;          // LDS row pitch of a staged source tile (floats)
constexpr int WS_HP = 68;          // LDS row pitch of the intermediate tile (16-byte writes from 16 rows at once stay conflict-free)
constexpr int WS_XS = WS_H * WS_XP + 64;   // floats per staged tile incl. the 16-float skew per 32 rows
constexpr int WS_HS = WS_H * WS_HP + 64;
constexpr int WS_MAX_JOBS = 8;
constexpr int WS_WG_PER_XCD = 64;  // 32 CUs x 2 resident workgroups

struct WsJob {
  const float* X;  long x_bs;      // source planes [B][128][P]
  const float* X2; long x2_bs;     // ws_lin with two sources: the second one
  const float* W1; const float* W2;   // mlp: first / second stage matrix;  lin: matrix of source 1 / source 2
  int ldw1, ldw2;
  const float* b1; const float* b2;   // mlp: stage biases; lin: b1 only (nullable)
  const float* Hsaved; long hs_bs;    // mlp backward: the saved hidden activations
  float* Mid; long mid_bs;            // mlp: where the intermediate is saved (SAVE_MID)
  float* Y; long y_bs;
  int accumulate;                     // Y += result
};

struct WsArgs {
  WsJob job[WS_MAX_JOBS];
  int njobs;
  const float* maskp;                 // [B][P] or null: multiplies the result
  // lin, optional rank-2 side input (the [A, sigma I] planes of layer_cat_lin): result += wside[ch][0..1] . side[b][0..1][p]
  const float* side; long side_bs;    // [B][2][P]
  const float* wside; int ldws;       // [128][ldws], columns 0 and 1
  int act;                            // lin: LeakyReLU on the result
  int B, P;
  // Instance norm folded into producer and consumers (split-bf16 inference kernels, dppm_wsplit.h; null / 0 elsewhere).
  //   aff:   [B][2][128] (scale, shift) of the masked instance norm of the source planes: the consumer stages x * scale + shift
  //          (fused MLP: its source X; two-source Linear: its second source X2, the skip input).  Masked positions are NOT zeroed:
  //          every consumer's result is masked or unobservable there.
  //   stat:  the two-source Linear also emits, per 64-position tile and output channel, the masked mean and the centred sum of
  //          squares of its result: stat[tile][128][2], cnt[tile] = number of unmasked positions of the tile (mask from maskstat
  //          [B][P]); norm_finalize_kernel merges the tiles of a graph into the next aff table.
  const float* aff;
  float* stat; float* cnt; const float* maskstat;
  // in_lin folded into layer 0's consumers (role-split kernels of dppm_w4.hip): when inl_w is set, the 128-channel source (fused MLP:
  // X; two-source Linear: X2) is not read but formed while staging, u0[c][p] = inl_w[c][0] * in[b][0][p] + inl_w[c][1] * in[b][1][p]
  // + inl_b[c] (model/ppgn.py:301), from the two input planes inl_in [B][2][P] (A and sigma I).
  const float* inl_w; const float* inl_b; const float* inl_in; long inl_bs;
  int dbg;                            // timing experiments (DPPM_DBG): bit 0 skips the statistics, bit 1 the affine staging
};

// one in_lin value; the stand-alone kernel and the staging code of the fused forms share this expression
// (explicit roundings: product, fused multiply-add, sum -- the order every form of it has to share to stay bit-identical)
__device__ __forceinline__ float ws_inlin(float w0, float w1, float bc, float a, float s) { return __fadd_rn(__fmaf_rn(w1, s, __fmul_rn(w0, a)), bc); }

__device__ __forceinline__ int ws_hoff(int k) { return k * WS_HP + 16 * (k >> 5); }

// a lane's 32 weights of output channel ch: w[s] = W[ch][32 g + s]   (TRANS: W^T, i.e. w[s] = W[32 g + s][ch])
template <bool TRANS>
__device__ __forceinline__ void ws_load_w(float (&w)[32], const float* __restrict__ W, int ldw, int ch, int g) {
  if (TRANS) {
#pragma unroll
    for (int s = 0; s < 32; ++s) w[s] = W[(size_t)(32 * g + s) * ldw + ch];
  } else {
    const float* p = W + (size_t)ch * ldw + 32 * g;
    if ((((uintptr_t)p) & 15) == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float4 q = reinterpret_cast<const float4*>(p)[v];
        w[4 * v + 0] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 32; ++s) w[s] = p[s];
    }
  }
}

// acc[j] (16 positions x 16 channels, j = 0..3) += T[:, 16 j ..]^T * Wslice^T ; T points at the lane's first element
template <int PITCH>
__device__ __forceinline__ void ws_mma4(floatx4 (&acc)[4], const float (&w)[32], const float* __restrict__ T) {
#pragma unroll
  for (int s = 0; s < 32; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[s * PITCH + 16 * j], w[s], acc[j], 0, 0, 0);
  }
}

// global tile (128 rows x 64 positions) -> registers (4 float4 per thread) / registers -> LDS.
// Xt: uniform pointer to (row 0, first position of the tile); o0 = (tid >> 4) * P + 4 * (tid & 15); ok: the lane's four
// positions lie inside the plane.  Uniform base + 32-bit lane offset keeps the address math off the vector registers.
__device__ __forceinline__ void ws_tile_load(float4 (&r)[4], const float* __restrict__ Xt, unsigned P, unsigned o0, bool ok) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float* Xv = Xt + (size_t)(32 * v) * P;
    r[v] = ok ? *reinterpret_cast<const float4*>(Xv + o0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ws_tile_store(const float4 (&r)[4], float* __restrict__ T, int tid) {
  float* q = T + (tid >> 4) * WS_XP + (tid & 15) * 4;
#pragma unroll
  for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(q + 32 * v * WS_XP + 16 * v) = r[v];   // row (tid >> 4) + 32 v, skew 16 v
}

struct WsWho { int jid, rank, per_job; };
__device__ __forceinline__ WsWho ws_who(int njobs) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  WsWho w;
  w.jid = __builtin_amdgcn_readfirstlane(slot % njobs);
  w.rank = __builtin_amdgcn_readfirstlane((slot / njobs) * 8 + xcd);
  w.per_job = __builtin_amdgcn_readfirstlane((int)(gridDim.x >> 3) / njobs * 8);
  return w;
}

__device__ __forceinline__ float ws_leaky(float v) { return fmaxf(v, kSlope * v); }   // = v > 0 ? v : slope v for 0 < slope < 1, one instruction less
__device__ __forceinline__ float4 ws_f4(const floatx4& a) { return make_float4(a[0], a[1], a[2], a[3]); }
// value * mask with 0 * anything = 0 (v_mul_legacy_f32): since the instance norm is folded into its consumers the layer tensors hold RAW
// m4 results, and a masked position is never zeroed between layers -- it evolves as rstd (W x + b) - mean rstd with rstd up to 1/sqrt(eps)
// on a near-constant channel, so after enough layers it may be Inf or NaN, and Inf * 0 = NaN in o1 / o2 would reach every logit of the
// graph through the contraction.  The reference zeroes these positions at every norm (model/ppgn.py:315).  For mask values in {0, 1}
// and finite values this is the plain product (up to the sign of a zero).
__device__ __forceinline__ float ws_mask_mul(float v, float m) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
  return r;
}
__device__ __forceinline__ void ws_mask_mul4(float4& v, const float4& m) { v.x = ws_mask_mul(v.x, m.x); v.y = ws_mask_mul(v.y, m.y); v.z = ws_mask_mul(v.z, m.z); v.w = ws_mask_mul(v.w, m.w); }

// result tile -> global: optional LeakyReLU, mask, accumulate.  Yt / Mt: uniform pointers to the tile's first position;
// r0 = ch * P + 4 g, m0 = 4 g: the lane's offsets; lim = positions left in the plane from the tile's start.
__device__ __forceinline__ void ws_store_out(const floatx4 (&acc)[4], float* __restrict__ Yt, unsigned r0, const float* __restrict__ Mt,
                                             unsigned m0, int lim, float bias, bool act, bool accumulate) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if ((int)m0 + 16 * j < lim) {
      float4 v = ws_f4(acc[j]);
      v.x += bias; v.y += bias; v.z += bias; v.w += bias;   // after the chain, as a GEMM epilogue would: one rounding at the bias' magnitude
      if (act) { v.x = ws_leaky(v.x); v.y = ws_leaky(v.y); v.z = ws_leaky(v.z); v.w = ws_leaky(v.w); }
      if (Mt) {
        const float4 m = *reinterpret_cast<const float4*>(Mt + m0 + 16 * j);
        ws_mask_mul4(v, m);
      }
      if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(Yt + r0 + 16 * j);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(Yt + r0 + 16 * j) = v;
    }
  }
}

// The same with WIDE stores.  In the MFMA result layout a wave's store instruction covers 16 channel rows x 64 bytes: half a
// 128-byte line per row and instruction, and the L2 sees every line twice (tools/dbg/cubw.hip: the MLP pair's traffic shape takes
// 0.49 ms with such stores against 0.39 ms with 256-byte row pieces).  A 4 x 4 transpose inside every quad of lanes (ci & 3) x
// (position groups j), two quad_perm DPP exchanges per register, turns the lane's data into: register k = channel 4 (ci >> 2) + k,
// positions 16 (ci & 3) + 4 g ..: an instruction then covers 4 rows x 256 contiguous bytes.
__device__ __forceinline__ float ws_dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float ws_dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ void ws_quad_transpose(floatx4 (&a)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int p = 0; p < 4; p += 2) {                         // lanes ci ^ 1 trade registers (p, p + 1)
      const float x = a[p][c], y = a[p + 1][c];
      const float recv = ws_dpp_xor1(b0 ? x : y);
      a[p][c] = b0 ? recv : x; a[p + 1][c] = b0 ? y : recv;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {                            // lanes ci ^ 2 trade registers (p, p + 2)
      const float x = a[p][c], y = a[p + 2][c];
      const float recv = ws_dpp_xor2(b1 ? x : y);
      a[p][c] = b1 ? recv : x; a[p + 2][c] = b1 ? y : recv;
    }
  }
}
// acc (destroyed): the lane's 16 results, channel ch0 + ci, positions 16 j + 4 g + r.  Yt: uniform pointer to the tile's first
// position of channel 0; bvp: null (bias already added) or ws_bias_wide(bias, ch0, lane), loaded once per kernel; Mt: the tile's mask row (LDS or global) or null.
__device__ __forceinline__ float4 ws_bias_wide(const float* __restrict__ bias, int ch0, int lane) {   // of channels ch0 + (ci & 12) + 0..3
  if (!bias) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float* b = bias + ch0 + (lane & 12);
  return make_float4(b[0], b[1], b[2], b[3]);
}
__device__ __forceinline__ void ws_store_wide(floatx4 (&acc)[4], float* __restrict__ Yt, int ch0, int P, const float4* bvp,
                                              const float* __restrict__ Mt, int lim, int lane, bool act, bool accumulate) {
  ws_quad_transpose(acc, lane);
  const int ci = lane & 15, g = lane >> 4, q = ci & 3, c4 = ch0 + (ci & 12);
  const int pos = 16 * q + 4 * g;
  if (pos >= lim) return;
  float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
  if (Mt) m = *reinterpret_cast<const float4*>(Mt + pos);
  float* Y = Yt + (size_t)c4 * P + pos;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 v = ws_f4(acc[k]);
    if (bvp) {
      const float bk = k == 0 ? bvp->x : (k == 1 ? bvp->y : (k == 2 ? bvp->z : bvp->w));
      v.x += bk; v.y += bk; v.z += bk; v.w += bk;
    }
    if (act) { v.x = ws_leaky(v.x); v.y = ws_leaky(v.y); v.z = ws_leaky(v.z); v.w = ws_leaky(v.w); }
    ws_mask_mul4(v, m);
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(Y + (size_t)k * P);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(Y + (size_t)k * P) = v;
  }
}

// The same transposition through a wave-private LDS patch R (16 rows of WS_WIDE_PITCH floats, 16-byte aligned) instead of the
// DPP exchanges: 8 LDS instructions in place of ~64 vector instructions per 16 x 64 result block, for kernels whose waves are short
// of vector issue slots rather than of LDS cycles.  bias: the lane's channel ch0 + ci.  The pitch keeps both the b128 stores (8
// lanes of one g, rows ci .. ci+7) and the b128 loads (16 lanes along a row) free of bank conflicts.
constexpr int WS_WIDE_PITCH = 68;
template <bool ADDB = true>   // ADDB = false: the accumulators already hold the bias
__device__ __forceinline__ void ws_store_wide_lds(const floatx4 (&acc)[4], float* __restrict__ R, float* __restrict__ Yt, int ch0, int P, float bias,
                                                  const float* __restrict__ Mt, int lim, int lane) {
  const int ci = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(R + ci * WS_WIDE_PITCH + 16 * j + 4 * g) =
        ADDB ? make_float4(acc[j][0] + bias, acc[j][1] + bias, acc[j][2] + bias, acc[j][3] + bias) : make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  const int pos = 4 * ci;
  float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
  if (Mt) m = *reinterpret_cast<const float4*>(Mt + pos);
  float* Y = Yt + (size_t)(ch0 + g) * P + pos;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 v = *reinterpret_cast<const float4*>(R + (4 * k + g) * WS_WIDE_PITCH + pos);
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;   // (plain product: the sources' masked positions are finite, see w4_lin2_kernel's store; ws_mask_mul here costs 4 %)
    if (pos < lim) *reinterpret_cast<float4*>(Y + (size_t)(4 * k) * P) = v;
  }
}

template <bool BWD, bool TRANS, bool SAVE_MID>
__global__ __launch_bounds__(WS_THREADS, 4) void ws_mlp_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Hs = smem + WS_XS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15, ch = 16 * wave + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;

  float w1[32], w2[32];
  ws_load_w<TRANS>(w1, J.W1, J.ldw1, ch, g);
  ws_load_w<TRANS>(w2, J.W2, J.ldw2, ch, g);
  const float b1v = J.b1 ? J.b1[ch] : 0.f, b2v = J.b2 ? J.b2[ch] : 0.f;

  const unsigned o0 = (unsigned)(tid >> 4) * P + 4 * (tid & 15);   // staging: row tid >> 4, 4 positions
  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;                    // results: row ch, positions 16 j + 4 g ..
  const unsigned m0 = 4 * g;

  float4 pre[4];
  int t = who.rank;
  if (t < ntiles) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    ws_tile_load(pre, J.X + (size_t)b * J.x_bs + p0, P, o0, p0 + oc < P);
    ws_tile_store(pre, Xs, tid);
  }
  __syncthreads();

  const float* xa = Xs + (32 * g) * WS_XP + 16 * g + ci;   // lane's A-operand base: row 32 g, position ci
  const float* ha = Hs + (32 * g) * WS_HP + 16 * g + ci;
  float* hw = Hs + ws_hoff(ch) + 4 * g;                    // lane's intermediate slot: row ch, positions 16 j + 4 g ..

  for (; t < ntiles; t += who.per_job) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const int tn = t + who.per_job;
    const int lim = P - p0;
    float4 hs[4];
    if (BWD) {
      const float* Ht = J.Hsaved + (size_t)b * J.hs_bs + p0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hs[j] = ((int)m0 + 16 * j < lim) ? *reinterpret_cast<const float4*>(Ht + r0 + 16 * j) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    ws_mma4<WS_XP>(acc, w1, xa);
    // ---- elementwise stage; intermediate to LDS (and to HBM when a later pass needs it)
    float* Mt = SAVE_MID ? J.Mid + (size_t)b * J.mid_bs + p0 : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v;
      if (BWD) {
        v.x = acc[j][0] * (hs[j].x > 0.f ? 1.f : kSlope); v.y = acc[j][1] * (hs[j].y > 0.f ? 1.f : kSlope);
        v.z = acc[j][2] * (hs[j].z > 0.f ? 1.f : kSlope); v.w = acc[j][3] * (hs[j].w > 0.f ? 1.f : kSlope);
      } else {
        v.x = ws_leaky(acc[j][0] + b1v); v.y = ws_leaky(acc[j][1] + b1v); v.z = ws_leaky(acc[j][2] + b1v); v.w = ws_leaky(acc[j][3] + b1v);
      }
      *reinterpret_cast<float4*>(hw + 16 * j) = v;
      if (SAVE_MID) {
        if ((int)m0 + 16 * j < lim) *reinterpret_cast<float4*>(Mt + r0 + 16 * j) = v;
      }
    }
    __syncthreads();
    if (tn < ntiles) {   // next tile: in flight during stage 2
      const int bn = __builtin_amdgcn_readfirstlane(tn / tpp), pn = __builtin_amdgcn_readfirstlane((tn % tpp) * WS_TP);
      ws_tile_load(pre, J.X + (size_t)bn * J.x_bs + pn, P, o0, pn + oc < P);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    ws_mma4<WS_HP>(acc, w2, ha);
    if (tn < ntiles) ws_tile_store(pre, Xs, tid);          // every wave left stage 1 (the only reader of Xs) at the barrier above
    ws_store_out(acc, J.Y + (size_t)b * J.y_bs + p0, r0, a.maskp ? a.maskp + (size_t)b * P + p0 : nullptr, m0, lim, b2v, false,
                 J.accumulate != 0);
    __syncthreads();
  }
}

template <int NSRC, bool TRANS>
__global__ __launch_bounds__(WS_THREADS, 4) void ws_lin_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15, ch = 16 * wave + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;

  float w[NSRC][32];
  ws_load_w<TRANS>(w[0], J.W1, J.ldw1, ch, g);
  if (NSRC == 2) ws_load_w<TRANS>(w[NSRC - 1], J.W2, J.ldw2, ch, g);
  const float bv = J.b1 ? J.b1[ch] : 0.f;

  const unsigned o0 = (unsigned)(tid >> 4) * P + 4 * (tid & 15);
  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;

  float4 pre[4];
  int t = who.rank;
  if (t < ntiles) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    ws_tile_load(pre, J.X + (size_t)b * J.x_bs + p0, P, o0, p0 + oc < P);
    ws_tile_store(pre, smem, tid);
  }
  __syncthreads();

  int cur = 0;
  const int lane_off = (32 * g) * WS_XP + 16 * g + ci;
  for (; t < ntiles; t += who.per_job) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const int tn = t + who.per_job;
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      // the tile that follows this one in the stream: the next source of the same positions, or source 0 of the next tile
      const bool last = (s == NSRC - 1);
      const bool more = last ? (tn < ntiles) : true;
      if (more) {
        if (last) {
          const int bn = __builtin_amdgcn_readfirstlane(tn / tpp), pn = __builtin_amdgcn_readfirstlane((tn % tpp) * WS_TP);
          ws_tile_load(pre, J.X + (size_t)bn * J.x_bs + pn, P, o0, pn + oc < P);
        } else {
          ws_tile_load(pre, J.X2 + (size_t)b * J.x2_bs + p0, P, o0, p0 + oc < P);
        }
      }
      ws_mma4<WS_XP>(acc, w[s], smem + cur * WS_XS + lane_off);
      if (more) ws_tile_store(pre, smem + (cur ^ 1) * WS_XS, tid);
      if (last) {
        if (a.side) {   // uniform branch
          const float s0 = a.wside[(size_t)ch * a.ldws], s1 = a.wside[(size_t)ch * a.ldws + 1];
          const float* St = a.side + (size_t)b * a.side_bs + p0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if ((int)m0 + 16 * j < P - p0) {
              const float4 x0 = *reinterpret_cast<const float4*>(St + m0 + 16 * j);
              const float4 x1 = *reinterpret_cast<const float4*>(St + P + m0 + 16 * j);
              acc[j][0] += s0 * x0.x + s1 * x1.x; acc[j][1] += s0 * x0.y + s1 * x1.y;
              acc[j][2] += s0 * x0.z + s1 * x1.z; acc[j][3] += s0 * x0.w + s1 * x1.w;
            }
          }
        }
        ws_store_out(acc, J.Y + (size_t)b * J.y_bs + p0, r0, a.maskp ? a.maskp + (size_t)b * P + p0 : nullptr, m0, P - p0, bv, a.act != 0,
                     J.accumulate != 0);
      }
      __syncthreads();
      cur ^= 1;
    }
  }
}

inline size_t ws_mlp_lds() { return (size_t)(WS_XS + WS_HS) * sizeof(float); }
inline size_t ws_lin_lds() { return (size_t)(2 * WS_XS) * sizeof(float); }

inline int ws_grid(int njobs, long ntiles) {
  long slots = WS_WG_PER_XCD / njobs;                 // workgroups per job and XCD
  const long need = (ntiles + 7) / 8;
  if (slots > need) slots = need;
  if (slots < 1) slots = 1;
  return (int)(8 * njobs * slots);
}

template <typename K>
inline int ws_launch_kernel(K kernel, size_t lds, const WsArgs& a, hipStream_t st, const char* what) {
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kernel), lds, what));
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  hipLaunchKernelGGL(kernel, dim3(ws_grid(a.njobs, ntiles)), dim3(WS_THREADS), lds, st, a);
  return check_launch(what);
}

inline bool ws_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline int ws_check(const WsArgs& a, const char* what) {
  if (a.P <= 0 || a.P % 4 != 0 || a.B <= 0) return fail(DPPM_E_INVALID, "%s: B=%d P=%d (P must be a positive multiple of 4)", what, a.B, a.P);
  if (a.njobs < 1 || a.njobs > WS_MAX_JOBS) return fail(DPPM_E_INVALID, "%s: njobs=%d", what, a.njobs);
  if (a.maskp && !ws_al16(a.maskp)) return fail(DPPM_E_INVALID, "%s: mask planes must be 16-byte aligned", what);
  for (int i = 0; i < a.njobs; ++i) {
    const WsJob& j = a.job[i];
    if (!j.X || !j.W1 || !j.Y) return fail(DPPM_E_INVALID, "%s: null pointer in job %d", what, i);
    if (!ws_al16(j.X) || !ws_al16(j.Y) || j.x_bs % 4 != 0 || j.y_bs % 4 != 0) return fail(DPPM_E_INVALID, "%s: planes must be 16-byte aligned", what);
  }
  return DPPM_OK;
}

// Algorithmic HBM bytes of one launch for the measurement hooks: every distinct source plane set read once, every result written
// once (read as well when accumulating), plus the saved / re-read hidden activations of the training forms.
inline double ws_alg_bytes(const WsArgs& a, int nsrc, bool save_mid, bool reads_hidden) {
  const float* seen[2 * WS_MAX_JOBS];
  int ns = 0;
  double sets = 0;
  auto src = [&](const float* p) {
    for (int i = 0; i < ns; ++i)
      if (seen[i] == p) return;
    seen[ns++] = p;
    sets += 1;
  };
  for (int i = 0; i < a.njobs; ++i) {
    src(a.job[i].X);
    if (nsrc == 2) src(a.job[i].X2);
    sets += 1 + (a.job[i].accumulate ? 1 : 0) + (save_mid ? 1 : 0) + (reads_hidden ? 1 : 0);
  }
  return sets * sizeof(float) * (double)WS_H * a.P * a.B;
}

// Linear -> LeakyReLU -> Linear (* mask) for every job; save_mid keeps the hidden activations (training).
inline int ws_mlp_forward(const WsArgs& a, bool save_mid, hipStream_t st) {
  DPPM_TRY(ws_check(a, "ws_mlp_forward"));
  for (int i = 0; i < a.njobs; ++i) {
    if (!a.job[i].W2) return fail(DPPM_E_INVALID, "ws_mlp_forward: job %d lacks its second matrix", i);
    if (save_mid && (!a.job[i].Mid || !ws_al16(a.job[i].Mid) || a.job[i].mid_bs % 4)) return fail(DPPM_E_INVALID, "ws_mlp_forward: job %d: bad Mid", i);
  }
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  const int rc = save_mid ? ws_launch_kernel(ws_mlp_kernel<false, false, true>, ws_mlp_lds(), a, st, "ws_mlp_forward(save)")
                          : ws_launch_kernel(ws_mlp_kernel<false, false, false>, ws_mlp_lds(), a, st, "ws_mlp_forward");
  if (prof) prof_after(st, PROF_WS_MLP2, AR_F32, 4.0 * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, 1, save_mid, false));
  return rc;
}

// dpre = (W1^T dOut) * LeakyReLU'(hid) -> Mid ; Y (+)= W2^T dpre      (W1 = the MLP's second matrix, W2 = its first)
inline int ws_mlp_backward(const WsArgs& a, hipStream_t st) {
  DPPM_TRY(ws_check(a, "ws_mlp_backward"));
  for (int i = 0; i < a.njobs; ++i)
    if (!a.job[i].W2 || !a.job[i].Hsaved || !a.job[i].Mid || !ws_al16(a.job[i].Hsaved) || !ws_al16(a.job[i].Mid) || a.job[i].hs_bs % 4 ||
        a.job[i].mid_bs % 4)
      return fail(DPPM_E_INVALID, "ws_mlp_backward: job %d needs W2 and 16-byte aligned Hsaved and Mid", i);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  const int rc = ws_launch_kernel(ws_mlp_kernel<true, true, true>, ws_mlp_lds(), a, st, "ws_mlp_backward");
  if (prof) prof_after(st, PROF_WS_MLP2, AR_F32, 4.0 * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, 1, true, true));
  return rc;
}

// Y (+)= epilogue(W1 X [+ W2 X2] + b1) for every job; trans applies the transposed matrices (data gradients).
inline int ws_lin(const WsArgs& a, int nsrc, bool trans, hipStream_t st) {
  DPPM_TRY(ws_check(a, "ws_lin"));
  if (a.side && (!a.wside || !ws_al16(a.side) || a.side_bs % 4)) return fail(DPPM_E_INVALID, "ws_lin: bad side input");
  if (nsrc == 2)
    for (int i = 0; i < a.njobs; ++i)
      if (!a.job[i].X2 || !a.job[i].W2 || !ws_al16(a.job[i].X2) || a.job[i].x2_bs % 4) return fail(DPPM_E_INVALID, "ws_lin: job %d lacks its second source", i);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (nsrc == 1) rc = trans ? ws_launch_kernel(ws_lin_kernel<1, true>, ws_lin_lds(), a, st, "ws_lin<1,T>")
                            : ws_launch_kernel(ws_lin_kernel<1, false>, ws_lin_lds(), a, st, "ws_lin<1,N>");
  else rc = trans ? ws_launch_kernel(ws_lin_kernel<2, true>, ws_lin_lds(), a, st, "ws_lin<2,T>")
                  : ws_launch_kernel(ws_lin_kernel<2, false>, ws_lin_lds(), a, st, "ws_lin<2,N>");
  if (prof) prof_after(st, nsrc == 2 ? PROF_WS_SUM2 : PROF_WS_COLS, AR_F32, 2.0 * nsrc * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, nsrc, false, false));
  return rc;
}

}  // namespace dppm
