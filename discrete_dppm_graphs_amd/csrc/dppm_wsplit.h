// Weight-stationary H = 128 linears on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 16x the f32 MFMA rate).
//
// Operands stay float32 in HBM.  While a tile is staged into LDS every value x is split into bf16 pieces
//   x = hi + lo + r,  hi = bf16(x) (round to nearest even),  lo = bf16(x - hi),  |r| <= 2^-17 |x|
// and a product is evaluated as  x w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w  (NS = 3; the dropped lo_x lo_w term is
// <= 2^-16 |x w|), accumulated in float32 by the MFMA.  A K = 128 Linear then agrees with the exact-f32 kernels of
// dppm_wstat.h to ~1e-5 of the result's magnitude, at 3/16 of their matrix time: the kernels become HBM-bound.
// NS = 1 keeps the hi pieces only (plain bf16 operands, f32 accumulate; tolerance stated in DESIGN.md).
//
// Same organisation as dppm_wstat.h (persistent workgroups, wave w owns output channels 16w .. 16w+15, weights
// register-resident for the whole launch, 64-position tiles) with these differences:
//   - a staged tile is an image [64 positions][128 k] of bf16 (k contiguous: a lane's MFMA fragment, 8 consecutive k
//     of one position, is one ds_read_b128), one image for hi and one for lo.  16-byte chunk c of position p lives at
//     chunk c ^ f(p), f(p) = (p & 15) ^ ((p >> 4) & 3): the fragment reads (16 lanes = 16 consecutive positions) and
//     the staging writes (16 lanes = positions 4 apart, 8 bytes each, two lanes per chunk) are both conflict-free;
//   - staging thread (q, cg) moves channels 4cg .. 4cg+3 of positions 4q .. 4q+3 (four float4 global loads of 256
//     contiguous bytes per 16 lanes) and writes, per position, 4 consecutive k as 8 bytes per image;
//   - in the fused MLP the first product is issued as D = W * X (lane gets 4 consecutive CHANNELS of one position =
//     8 bytes of the hidden image), the second as D = H^T * W^T (lane gets 4 consecutive POSITIONS of one channel =
//     one 16-byte global store), so neither result needs a transpose.
#pragma once
#include "dppm_wstat.h"

namespace dppm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int W3_IMG = WS_TP * WS_H * 2;   // bytes of one bf16 image of a tile (16 KB)

__device__ __forceinline__ unsigned w3_pack(float a, float b) {   // (bf16(a), bf16(b)) -> one dword, a in the low half
  const floatx2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <int NS>
__device__ __forceinline__ void w3_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = w3_pack(x0, x1);
  if (NS > 1) lo = w3_pack(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

__device__ __forceinline__ floatx4 w3_mfma(const uintx4& a, const uintx4& b, const floatx4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// a lane's weights of output channel ch as four K = 32 fragments: element e of fragment s is W[ch][32 s + 8 g + e]
// (TRANS: W[32 s + 8 g + e][ch])
template <int NS, bool TRANS>
__device__ __forceinline__ void w3_load_w(uintx4 (&wh)[4], uintx4 (&wl)[4], const float* __restrict__ W, int ldw, int ch, int g) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
    if (TRANS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = W[(size_t)(32 * s + 8 * g + e) * ldw + ch];
    } else {
      const float* p = W + (size_t)ch * ldw + 32 * s + 8 * g;
      if ((((uintptr_t)p) & 15) == 0) {
        const float4 q0 = reinterpret_cast<const float4*>(p)[0], q1 = reinterpret_cast<const float4*>(p)[1];
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h, l = 0;
      w3_split2<NS>(v[2 * e], v[2 * e + 1], h, l);
      wh[s][e] = h;
      wl[s][e] = l;
    }
  }
}

// global tile (128 channels x 64 positions) -> registers: thread (q = tid & 15, cg = tid >> 4) takes channels 4cg + v,
// positions 4q .. 4q+3.  o0 = 4 (tid >> 4) * P + 4 (tid & 15).
__device__ __forceinline__ void w3_tile_load(float4 (&r)[4], const float* __restrict__ Xt, unsigned P, unsigned o0, bool ok) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float* Xv = Xt + (size_t)v * P;
    r[v] = ok ? *reinterpret_cast<const float4*>(Xv + o0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// registers -> the bf16 image(s) of the tile
template <int NS>
__device__ __forceinline__ void w3_put4(char* __restrict__ T, unsigned off, float x0, float x1, float x2, float x3) {
  unsigned h01, l01 = 0, h23, l23 = 0;
  w3_split2<NS>(x0, x1, h01, l01);
  w3_split2<NS>(x2, x3, h23, l23);
  *reinterpret_cast<uint2*>(T + off) = make_uint2(h01, h23);
  if (NS > 1) *reinterpret_cast<uint2*>(T + W3_IMG + off) = make_uint2(l01, l23);
}
// thread (q, cg): channels 4 cg .. 4 cg + 3 of positions 4 q .. 4 q + 3
// -DW3_NATPAIR=1 (an experiment, off): the remainders x - hi formed in the registers' pairing (two POSITIONS of one channel: the halves
// of a loaded float4 are aligned register pairs) with one v_pk_add_f32 per two values, which saves the v_mov per pair that hipcc pays
// for packing the scalar form across float4s (33 of a staging wave's ~190 vector instructions per tile).  Bit-identical by
// construction and measured SLOWER in the MLP pair (0.5025 against 0.4937 ms with the packed first-product epilogue, W4_PKEPI); and
// with the values in that pairing hipcc turns the norm's x * scale + shift in head3_kernel's stagers into v_pk_fma_f32 with
// broadcast operand halves (op_sel:[0,1,1], op_sel_hi:[1,0,0]) -- that build is grossly nondeterministic (thousands of logits per
// forward), the third sighting of the packed operand-select effect and its most reproducible one.
#ifndef W3_NATPAIR
#define W3_NATPAIR 0
#endif
__device__ __forceinline__ floatx2 w3_sub2(floatx2 x, unsigned h0, unsigned h1, bool high) {
  floatx2 h = high ? floatx2{__uint_as_float(h0 & 0xffff0000u), __uint_as_float(h1 & 0xffff0000u)}
                   : floatx2{__uint_as_float(h0 << 16), __uint_as_float(h1 << 16)};
  floatx2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(h));
  return d;
}
template <int NS>
__device__ __forceinline__ void w3_tile_store4_qc(float4 r0, float4 r1, float4 r2, float4 r3, char* __restrict__ T, unsigned q, unsigned cg) {
  const unsigned base = (4 * q) * 256 + (cg & 1) * 8;
  const unsigned kc = (cg >> 1) ^ (4 * (q & 3)) ^ (q >> 2);   // swizzled chunk of position 4q; position 4q + e: kc ^ e
  if (NS == 1) {   // the single-piece form: without this the SLP vectoriser transposes the 4 x 4 block through a stack array (384 B / lane
                   // of scratch traffic per tile in w4_mlp_kernel<1,..> and head3_kernel<1,..>)
    asm("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w)); asm("" : "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
    asm("" : "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w)); asm("" : "+v"(r3.x), "+v"(r3.y), "+v"(r3.z), "+v"(r3.w));
    w3_put4<NS>(T, base + ((kc ^ 0) << 4), r0.x, r1.x, r2.x, r3.x);
    w3_put4<NS>(T, base + 256 + ((kc ^ 1) << 4), r0.y, r1.y, r2.y, r3.y);
    w3_put4<NS>(T, base + 512 + ((kc ^ 2) << 4), r0.z, r1.z, r2.z, r3.z);
    w3_put4<NS>(T, base + 768 + ((kc ^ 3) << 4), r0.w, r1.w, r2.w, r3.w);
    return;
  }
#if W3_NATPAIR
  // hi pieces, position e: (channels 0, 1) and (channels 2, 3)
  const unsigned ha0 = w3_pack(r0.x, r1.x), hb0 = w3_pack(r2.x, r3.x), ha1 = w3_pack(r0.y, r1.y), hb1 = w3_pack(r2.y, r3.y);
  const unsigned ha2 = w3_pack(r0.z, r1.z), hb2 = w3_pack(r2.z, r3.z), ha3 = w3_pack(r0.w, r1.w), hb3 = w3_pack(r2.w, r3.w);
  // remainders, channel v: positions (0, 1) and (2, 3)
  const floatx2 d0a = w3_sub2(floatx2{r0.x, r0.y}, ha0, ha1, false), d0b = w3_sub2(floatx2{r0.z, r0.w}, ha2, ha3, false);
  const floatx2 d1a = w3_sub2(floatx2{r1.x, r1.y}, ha0, ha1, true), d1b = w3_sub2(floatx2{r1.z, r1.w}, ha2, ha3, true);
  const floatx2 d2a = w3_sub2(floatx2{r2.x, r2.y}, hb0, hb1, false), d2b = w3_sub2(floatx2{r2.z, r2.w}, hb2, hb3, false);
  const floatx2 d3a = w3_sub2(floatx2{r3.x, r3.y}, hb0, hb1, true), d3b = w3_sub2(floatx2{r3.z, r3.w}, hb2, hb3, true);
  const unsigned o0 = base + ((kc ^ 0) << 4), o1 = base + 256 + ((kc ^ 1) << 4), o2 = base + 512 + ((kc ^ 2) << 4), o3 = base + 768 + ((kc ^ 3) << 4);
  *reinterpret_cast<uint2*>(T + o0) = make_uint2(ha0, hb0);
  *reinterpret_cast<uint2*>(T + o1) = make_uint2(ha1, hb1);
  *reinterpret_cast<uint2*>(T + o2) = make_uint2(ha2, hb2);
  *reinterpret_cast<uint2*>(T + o3) = make_uint2(ha3, hb3);
  *reinterpret_cast<uint2*>(T + W3_IMG + o0) = make_uint2(w3_pack(d0a[0], d1a[0]), w3_pack(d2a[0], d3a[0]));
  *reinterpret_cast<uint2*>(T + W3_IMG + o1) = make_uint2(w3_pack(d0a[1], d1a[1]), w3_pack(d2a[1], d3a[1]));
  *reinterpret_cast<uint2*>(T + W3_IMG + o2) = make_uint2(w3_pack(d0b[0], d1b[0]), w3_pack(d2b[0], d3b[0]));
  *reinterpret_cast<uint2*>(T + W3_IMG + o3) = make_uint2(w3_pack(d0b[1], d1b[1]), w3_pack(d2b[1], d3b[1]));
#else
  w3_put4<NS>(T, base + ((kc ^ 0) << 4), r0.x, r1.x, r2.x, r3.x);
  w3_put4<NS>(T, base + 256 + ((kc ^ 1) << 4), r0.y, r1.y, r2.y, r3.y);
  w3_put4<NS>(T, base + 512 + ((kc ^ 2) << 4), r0.z, r1.z, r2.z, r3.z);
  w3_put4<NS>(T, base + 768 + ((kc ^ 3) << 4), r0.w, r1.w, r2.w, r3.w);
#endif
}
template <int NS>
__device__ __forceinline__ void w3_tile_store4(float4 r0, float4 r1, float4 r2, float4 r3, char* __restrict__ T, int tid) {
  w3_tile_store4_qc<NS>(r0, r1, r2, r3, T, tid & 15, tid >> 4);
}
template <int NS>
__device__ __forceinline__ void w3_tile_store(const float4 (&r)[4], char* __restrict__ T, int tid) {
  w3_tile_store4<NS>(r[0], r[1], r[2], r[3], T, tid);
}
// one tile in flight: plain members (no array) so that the set provably stays in registers
struct W3Pre { float4 a, b, c, d; float m; };

// Tiles of a workgroup.  Default: rank, rank + per_job, ... (neighbouring workgroups stream neighbouring tiles).  CONTIG: one
// contiguous run per workgroup, so that it crosses a graph boundary at most once per tpp tiles and the per-graph normalisation
// tables of its whole run (W3_TAB graphs at most, checked on the host) sit in LDS for the launch.
constexpr int W3_TAB = 4;
struct W3Tiles { int first, stride, nloc; };
template <bool CONTIG>
__device__ __forceinline__ W3Tiles w3_tiles(const WsWho& who, int ntiles) {
  W3Tiles t;
  if (CONTIG) {
    const int chunk = (ntiles + who.per_job - 1) / who.per_job;
    t.first = who.rank * chunk; t.stride = 1;
    const int left = ntiles - t.first;
    t.nloc = left < 0 ? 0 : (left < chunk ? left : chunk);
  } else {
    t.first = who.rank; t.stride = who.per_job;
    t.nloc = who.rank < ntiles ? (ntiles - who.rank + who.per_job - 1) / who.per_job : 0;
  }
  return t;
}
// host: does every workgroup's contiguous run stay within W3_TAB graphs?
inline bool w3_contig_ok(long ntiles, int tpp, int per_job) {
  const long chunk = (ntiles + per_job - 1) / per_job;
  return (chunk + tpp - 1) / tpp + 1 <= W3_TAB;
}
// the (scale, shift) tables of graphs b0 .. b0 + W3_TAB - 1 -> LDS (1024 threads: one float each)
__device__ __forceinline__ void w3_load_tab(float* __restrict__ Tb, const float* __restrict__ aff, int b0, int B, int tid) {
  if (tid < W3_TAB * 256) {
    int b = b0 + (tid >> 8);
    if (b > B - 1) b = B - 1;
    Tb[tid] = aff[(size_t)b * 256 + (tid & 255)];
  }
}
// x * scale + shift on four positions of one channel.  The empty asm hides from the compiler that neighbouring channels' (scale, shift)
// sit in neighbouring registers of one LDS read: knowing it, hipcc broadcasts the HIGH register of such a pair into both halves of a
// v_pk_fma_f32 (op_sel:[0,1,1]) -- the operand-select form that computes with the wrong register from time to time on this part
// (DESIGN.md section 4); a scalar of unknown origin is broadcast from the LOW half (op_sel_hi:[1,0,0]), which never failed.
// OPAQUE is set where hipcc was seen to do that (w4_lin2_kernel<3, true, false>'s stagers; everywhere, it costs w4_mlp_kernel 16 spilled
// registers); tests/test_isa_gate.py fails the build on any such instruction in the default inference kernels, whoever produced it.
template <bool OPAQUE = false>
__device__ __forceinline__ void w3_affine(float4& v, float sc, float sh) {
  if (OPAQUE) asm("" : "+v"(sc), "+v"(sh));
  v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
}

// per-lane byte offsets of the fragment (position 16 j + ci, k = 32 s + 8 g ..): aj[j] ^ (s << 6)
// (chunk (4s | g) ^ ci ^ j: bits 2-3 are s ^ (ci >> 2), folded into aj as (ci >> 2) << 6, so s enters as one XOR)
struct W3Addr { unsigned aj[4]; };
__device__ __forceinline__ W3Addr w3_addr(int ci, int g) {
  W3Addr a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a.aj[j] = (16 * j + ci) * 256 + ((ci >> 2) << 6) + ((g ^ (ci & 3) ^ j) << 4);
  return a;
}
__device__ __forceinline__ unsigned w3_off(const W3Addr& a, int j, int s) { return a.aj[j] ^ (unsigned)(s << 6); }

// w0 a + w1 s + bc with ws_inlin's three roundings, as single-lane instructions that the compiler cannot pack.  Two kernels gave wrong
// values from run to run while hipcc had turned this expression into v_pk_mul_f32 / v_pk_fma_f32 with crossed operand halves
// (op_sel:[0,1] op_sel_hi:[1,0]) in waves that keep vector-memory loads in flight beside other waves' MFMAs: the in_lin rows of
// w4_mlp_kernel's stagers (operands: destination pairs of global loads) and the side-input term of head3_kernel's epilogue (operands:
// LDS reads, the next weight fragments in flight; round 2's global-load form of it was only deterministic behind a vmcnt(0)).
// Every build of either kernel without those packed forms is bit-exact and deterministic.  Reproduced outside the library
// (tools/dbg/pk_opsel_probe3.hip): a packed-f32 instruction with an op_sel bit set loses the product of its low half -- the result is the
// addend alone -- now and then while the wave has loads in flight and another wave of the SIMD runs MFMAs.
// -DW3_SIDE_PACKED gives the expression back to the compiler for that experiment (tools/dbg/head_packed_ab.sh).
__device__ __forceinline__ float w3_side_s(float w0, float w1, float bc, float a, float s) {
#ifdef W3_SIDE_PACKED
  return __fadd_rn(__fmaf_rn(w1, s, __fmul_rn(w0, a)), bc);
#endif
  float t;
  asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(w0), "v"(a));
  asm("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(w1), "v"(s));
  asm("v_add_f32 %0, %0, %1" : "+v"(t) : "v"(bc));
  return t;
}

// Staging thread t -> (position quad q, channel group cg) of the role-split kernels.  The 16 lanes of a ds_write_b64 group are
// served in one LDS cycle only if they cover 16 distinct 8-byte bank pairs of the 128-byte store row.  An image slot is (8-byte half
// cg & 1) of (16-byte chunk (cg >> 1) ^ 4 (q & 3) ^ (q >> 2) ^ e): with the plain map (q = t & 15, cg = t >> 4) the 16 lanes share cg,
// sit on one half and reach 8 chunks twice -- two cycles per group, a quarter of w4_mlp_kernel's LDS cycles (SQ_LDS_BANK_CONFLICT,
// profiles/r02f_pmc_mfma_lds.txt).  Here lane bit 0 is cg's low bit and q's bit 1 (which the chunk's low three bits do not see) is
// lane bit 4: 2 halves x 8 chunks.  A wave still covers four channel rows x 256 contiguous bytes per global load instruction.
// (Measured: the conflict cycles go, the launch time does not move -- the stores' LDS cycles were never on the critical path.)
__device__ __forceinline__ unsigned w4_q(int t) { return ((t >> 1) & 1) | (((t >> 4) & 1) << 1) | (((t >> 2) & 3) << 2); }
__device__ __forceinline__ unsigned w4_cg(int t) { return (t & 1) | ((t >> 5) << 1); }


// acc[j] += X-tile(positions 16 j ..) x W-slice over K = 128.  W_IS_A: D = W X (lane: 4 channels of one position);
// else D = X^T W^T (lane: 4 positions of one channel).
template <int NS, bool W_IS_A>
__device__ __forceinline__ void w3_mma4(floatx4 (&acc)[4], const uintx4 (&wh)[4], const uintx4 (&wl)[4], const char* __restrict__ T,
                                        const W3Addr& ad) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = w3_off(ad, j, s);
      const uintx4 xh = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) {
        const uintx4 xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
        acc[j] = W_IS_A ? w3_mfma(wh[s], xl, acc[j]) : w3_mfma(xl, wh[s], acc[j]);
        acc[j] = W_IS_A ? w3_mfma(wl[s], xh, acc[j]) : w3_mfma(xh, wl[s], acc[j]);
      }
      acc[j] = W_IS_A ? w3_mfma(wh[s], xh, acc[j]) : w3_mfma(xh, wh[s], acc[j]);
    }
  }
}

// Pivot of the one-pass tile statistics: the channel's value at the tile's FIRST UNMASKED position (mask row Mb[64]).  Lane (g, ci)
// holds positions 16 j + 4 g + r of channel ci in acc[j][r].  The pivot has to lie inside the data: a masked position holds
// anything, and a pivot k standard deviations off the mean costs S2 - S1^2 / n about eps k^2 of its digits.
__device__ __forceinline__ float w3_pivot(const floatx4 (&acc)[4], const float* Mb, int lane, int ci, float bv) {
  const unsigned long long bal = __ballot(Mb[lane] > 0.f);
  const int ps = __builtin_amdgcn_readfirstlane(bal ? __builtin_ctzll(bal) : 0);
  const int js = ps >> 4, gs = (ps >> 2) & 3, rs = ps & 3;
  float cand = acc[0][0];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cand = (js == j && rs == r) ? acc[j][r] : cand;
  return __shfl(cand, 16 * gs + ci, 64) + bv;
}

// Y (+)= epilogue(sum_s W_s X_s + b): one or two 128-channel sources, optional rank-2 side input, LeakyReLU, mask.
template <int NS, int NSRC, bool TRANS>
__global__ __launch_bounds__(WS_THREADS, 4) void w3_lin_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15, ch = 16 * wave + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;

  uintx4 wh[NSRC][4], wl[NSRC][4];
  w3_load_w<NS, TRANS>(wh[0], wl[0], J.W1, J.ldw1, ch, g);
  if (NSRC == 2) w3_load_w<NS, TRANS>(wh[NSRC - 1], wl[NSRC - 1], J.W2, J.ldw2, ch, g);
  const float bv = J.b1 ? J.b1[ch] : 0.f;

  const unsigned o0 = (unsigned)(4 * (tid >> 4)) * P + 4 * (tid & 15);
  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  const W3Addr ad = w3_addr(ci, g);

  // prefetch set in named registers; loads are unconditional (clamped) so that the compiler's counted waits stay exact
  W3Pre pre;
  auto issue = [&](const float* __restrict__ Xt, bool in) {
    const float* p = Xt + (in ? o0 : 0u);
    pre.a = *reinterpret_cast<const float4*>(p);
    pre.b = *reinterpret_cast<const float4*>(p + (size_t)P);
    pre.c = *reinterpret_cast<const float4*>(p + (size_t)2 * P);
    pre.d = *reinterpret_cast<const float4*>(p + (size_t)3 * P);
  };
  auto land = [&](char* T, bool in) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    w3_tile_store4<NS>(in ? pre.a : z, in ? pre.b : z, in ? pre.c : z, in ? pre.d : z, T, tid);
  };
  int t = who.rank;
  if (t >= ntiles) return;   // whole workgroup, before any barrier
  {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    issue(J.X + (size_t)b * J.x_bs + p0, p0 + oc < P);
    land(smem3, p0 + oc < P);
  }
  __syncthreads();

  int cur = 0;
  for (; t < ntiles; t += who.per_job) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const int tn = t + who.per_job < ntiles ? t + who.per_job : t;   // past the end: re-read this tile (never consumed)
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      // the tile that follows this one in the stream: the next source of the same positions, or source 0 of the next tile
      const bool last = (s == NSRC - 1);
      bool in;
      if (last) {
        const int bn = __builtin_amdgcn_readfirstlane(tn / tpp), pn = __builtin_amdgcn_readfirstlane((tn % tpp) * WS_TP);
        in = pn + oc < P;
        issue(J.X + (size_t)bn * J.x_bs + pn, in);
      } else {
        in = p0 + oc < P;
        issue(J.X2 + (size_t)b * J.x2_bs + p0, in);
      }
      w3_mma4<NS, false>(acc, wh[s], wl[s], smem3 + cur * NIMG * W3_IMG, ad);
      land(smem3 + (cur ^ 1) * NIMG * W3_IMG, in);
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


// ---------------------------------------------------------------------------------------------------------
// Fused MLP  out = (W2 LeakyReLU(W1 X + b1) + b2) * mask  (model/ppgn.py:58-59), inference form (the hidden tile never
// leaves LDS).  One 16-wave workgroup per CU with the two stages on DIFFERENT waves: waves 0-7 hold W1, compute the
// hidden tile of tile i (X image -> H image) and stage tile i+1 (global loads issued before the MFMAs of tile i, landed
// after them); waves 8-15 hold W2 and turn the hidden tile of tile i-1 into the output.  One barrier per tile; X and H
// images are double-buffered so neither stage waits for the other.
// Why the roles are split: with both stages on one wave (as in dppm_wstat.h) the bf16 stages are too short to cover the
// prefetch and store latencies, and a wave that both prefetches and stores must drain its fresh stores at every counted
// wait for the prefetch (one in-order counter for loads and stores).  Here the loading waves never store and the
// storing waves never wait: measured 1.39 ms (f32) -> 0.84 (one wave, both stages) -> 0.48 ms per MLP pair launch.
// ---------------------------------------------------------------------------------------------------------
constexpr int W3M_THREADS = 1024;

// two position groups (16 jb + ci, 16 (jb+1) + ci) x the wave's 16 channels over K = 128
template <int NS, bool W_IS_A>
__device__ __forceinline__ void w3_mma2(floatx4 (&acc)[2], const uintx4 (&wh)[4], const uintx4 (&wl)[4], const char* __restrict__ T,
                                        const W3Addr& ad, int jb) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = w3_off(ad, jb + j, s);
      const uintx4 xh = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) {
        const uintx4 xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
        acc[j] = W_IS_A ? w3_mfma(wh[s], xl, acc[j]) : w3_mfma(xl, wh[s], acc[j]);
        acc[j] = W_IS_A ? w3_mfma(wl[s], xh, acc[j]) : w3_mfma(xh, wl[s], acc[j]);
      }
      acc[j] = W_IS_A ? w3_mfma(wh[s], xh, acc[j]) : w3_mfma(xh, wh[s], acc[j]);
    }
  }
}

// DEPTH = tiles in flight per staging wave (1: 16 registers, 2: 32): the short bf16 stages need 2, the 3-MFMA stages of the
// split form cover the latency with 1 (and have no registers to spare).
template <int NS, int DEPTH, bool NRM>
__global__ __launch_bounds__(W3M_THREADS) void w3_mlp_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  char* Xs = smem3;                       // [2][NIMG] images
  char* Hs = smem3 + 2 * BUF;             // [2][NIMG] images
  float* Ms = reinterpret_cast<float*>(smem3 + 4 * BUF);   // [4][64] mask rows: tile i is written in iteration i-1 and read in i+1
  float* Tb = Ms + 4 * 64;                                 // NRM: [W3_TAB][2][128] (scale, shift) of the graphs this workgroup meets
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w8 + ci;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  const W3Tiles tl = w3_tiles<NRM>(who, ntiles);
  if (tl.nloc <= 0) return;   // whole workgroup, before any barrier
  const int b0 = tl.first / tpp;
  if (NRM) {                                              // the prologue's first landing already reads the table
    w3_load_tab(Tb, a.aff, b0, a.B, tid);
    __syncthreads();
  }

  uintx4 wh[4], wl[4];
  w3_load_w<NS, false>(wh, wl, role ? J.W2 : J.W1, role ? J.ldw2 : J.ldw1, ch, g);
  float bv[4];   // stage 1: biases of channels 16 w8 + 4 g + r; stage 2: bias of channel ch in bv[0]
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = role ? (J.b2 ? J.b2[ch] : 0.f) : (J.b1 ? J.b1[16 * w8 + 4 * g + r] : 0.f);

  const unsigned o0 = (unsigned)(4 * ((tid & 511) >> 4)) * P + 4 * (tid & 15);   // staging (waves 0-7): channels 4cg.., positions 4q..
  const int oc = 4 * (tid & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  const W3Addr ad = w3_addr(ci, g);
  // stage 1 result slot in the hidden image: position 16 j + ci, k = 16 w8 + 4 g .. +3 (chunk 2 w8 + (g >> 1), half g & 1)
  const unsigned hbase = ci * 256 + (g & 1) * 8, hchunk = (2 * w8 + (g >> 1)) ^ ci;

  // tiles of this workgroup: t_i = first + i * stride, i < nloc (>= 1).
  // Every global load below is UNCONDITIONAL straight-line code (indices clamped instead of predicated): a load inside
  // a lane- or loop-dependent branch makes the compiler drain the whole queue (s_waitcnt vmcnt(0)) at the join, which
  // serialises the prefetch with the compute.
  const int nloc = tl.nloc;
  const float* maskp = a.maskp;
  // Only the stage-1 waves stage tiles: their queue then holds loads only, so the counted waits the compiler emits are
  // exact.  (A wave that also stores would have to drain its just-issued stores before every landing.)
  auto issue = [&](int i, W3Pre& pre) {   // global loads of tile min(i, nloc-1) into a prefetch register set
    const int t = tl.first + (i < nloc ? i : nloc - 1) * tl.stride;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const float* Xt = J.X + (size_t)b * J.x_bs + p0 + ((p0 + oc < P) ? o0 : 0u);   // ragged last tile: re-read in-range data, zeroed when landed
    pre.a = *reinterpret_cast<const float4*>(Xt);
    pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
    pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
    pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
    if (maskp) pre.m = maskp[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
  };
  auto land = [&](int i, const W3Pre& pre) {   // register set -> images of buffer i & 1, mask row i & 3
    const int t = tl.first + (i < nloc ? i : nloc - 1) * tl.stride;
    const int p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const bool in = p0 + oc < P;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xa = in ? pre.a : z, xb = in ? pre.b : z, xc = in ? pre.c : z, xd = in ? pre.d : z;
    if (NRM) {   // the source's instance norm: channel 4 cg + v of graph b
      const float* T = Tb + (t / tpp - b0) * 256 + 4 * (tid >> 4);
      const float4 sc = *reinterpret_cast<const float4*>(T), sh = *reinterpret_cast<const float4*>(T + 128);
      w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
    }
    w3_tile_store4<NS>(xa, xb, xc, xd, Xs + (i & 1) * BUF, tid);
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
        for (int jb = 0; jb < 4; jb += 2) {
          floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
          w3_mma2<NS, true>(acc, wh, wl, Xs + (i & 1) * BUF, ad, jb);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unsigned h01, l01 = 0, h23, l23 = 0;
            w3_split2<NS>(ws_leaky(acc[j][0] + bv[0]), ws_leaky(acc[j][1] + bv[1]), h01, l01);
            w3_split2<NS>(ws_leaky(acc[j][2] + bv[2]), ws_leaky(acc[j][3] + bv[3]), h23, l23);
            const unsigned off = hbase + (jb + j) * 4096 + ((hchunk ^ (jb + j)) << 4);
            *reinterpret_cast<uint2*>(Hb + off) = make_uint2(h01, h23);
            if (NS > 1) *reinterpret_cast<uint2*>(Hb + W3_IMG + off) = make_uint2(l01, l23);
          }
        }
      }
      land(i + 1, pl);
    } else if (i >= 1) {
      const int t = tl.first + (i - 1) * tl.stride;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
      const float* Mb = Ms + ((i - 1) & 3) * 64;
      float* Yt = J.Y + (size_t)b * J.y_bs + p0;
      const int lim = P - p0;
#pragma unroll
      for (int jb = 0; jb < 4; jb += 2) {
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        w3_mma2<NS, false>(acc, wh, wl, Hs + ((i - 1) & 1) * BUF, ad, jb);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int pj = (int)m0 + 16 * (jb + j);
          if (pj < lim) {
            float4 v = ws_f4(acc[j]);
            v.x += bv[0]; v.y += bv[0]; v.z += bv[0]; v.w += bv[0];
            if (maskp) {
              const float4 m = *reinterpret_cast<const float4*>(Mb + pj);
              ws_mask_mul4(v, m);
            }
            *reinterpret_cast<float4*>(Yt + r0 + 16 * (jb + j)) = v;
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

template <int NS> inline size_t w3m_lds() { return (size_t)(NS > 1 ? 8 : 4) * W3_IMG + 4 * 64 * sizeof(float) + W3_TAB * 256 * sizeof(float); }
inline long w3_slots(int njobs, long ntiles) {   // workgroups per job and XCD (one workgroup per CU: 32 per XCD)
  long slots = 32 / njobs;
  const long need = (ntiles + 7) / 8;
  if (slots > need) slots = need;
  return slots < 1 ? 1 : slots;
}

template <int NS, int DEPTH = (NS > 1 ? 1 : 2)>
inline int w3_mlp_forward(const WsArgs& a, hipStream_t st) {
  DPPM_TRY(ws_check(a, "w3_mlp_forward"));
  for (int i = 0; i < a.njobs; ++i) {
    if (!a.job[i].W2) return fail(DPPM_E_INVALID, "w3_mlp_forward: job %d lacks its second matrix", i);
    if (a.job[i].accumulate) return fail(DPPM_E_UNSUPPORTED, "w3_mlp_forward: accumulate is not provided");
  }
  const size_t lds = w3m_lds<NS>();
  const bool nrm = a.aff != nullptr;
  auto kern = nrm ? w3_mlp_kernel<NS, DEPTH, true> : w3_mlp_kernel<NS, DEPTH, false>;
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds), "w3_mlp_forward"));
  const int tpp = (a.P + WS_TP - 1) / WS_TP;
  const long ntiles = (long)a.B * tpp;
  const long slots = w3_slots(a.njobs, ntiles);
  if (nrm && !w3_contig_ok(ntiles, tpp, (int)(8 * slots))) return fail(DPPM_E_UNSUPPORTED, "w3_mlp_forward: normalisation tables of a workgroup's tile run exceed LDS");
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.njobs * slots)), dim3(W3M_THREADS), lds, st, a);
  const int rc = check_launch("w3_mlp_forward");
  if (prof) prof_after(st, PROF_WS_MLP2, prof_arith_of_pieces(NS), 4.0 * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, 1, false, false));
  return rc;
}


// ---------------------------------------------------------------------------------------------------------
// Two-source Linear  Y (+)= epilogue(W1 X1 + W2 X2 + b)  (the skip-concat m4 of model/ppgn.py:66-69, layer_cat_lin two
// layer outputs at a time), split form.  Holding both matrices' hi and lo pieces (64 registers) next to the accumulators
// and a prefetch set does not fit 128 registers, so the SOURCES go to different waves of one 16-wave workgroup per CU:
// waves 0-7 hold W1 and stage / multiply X1, waves 8-15 hold W2 and stage / multiply X2; the second group hands its
// partial tile to the first through LDS (lane-linear float4 rows: conflict-free both ways), which adds, applies the
// epilogue and stores.  Per tile: MFMAs, barrier, [land tile t+1 (requested a whole tile earlier), request tile t+2,
// epilogue of tile t], barrier.
// ---------------------------------------------------------------------------------------------------------
// NRM: the second source is staged through its instance norm (a.aff) and the result's per-tile statistics are emitted (a.stat,
// a.cnt; see WsArgs).
template <int NS, bool NRM>
__global__ __launch_bounds__(W3M_THREADS) void w3_lin2_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w8 + ci;
  char* Xs = smem3 + role * BUF;                                         // this group's source images
  float4* Ps = reinterpret_cast<float4*>(smem3 + 2 * BUF) + (w8 * 4) * 64 + lane;   // partial tile: [w8][j][lane] float4
  float* Ms = reinterpret_cast<float*>(smem3 + 2 * BUF + WS_H * WS_TP * sizeof(float));   // NRM: [2][64] mask rows of tiles i, i+1
  float* Tb = Ms + 2 * 64;                                                              // NRM: [W3_TAB][2][128] (scale, shift)
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  const W3Tiles tl = w3_tiles<NRM>(who, ntiles);
  if (tl.nloc <= 0) return;   // whole workgroup, before any barrier
  const int b0 = tl.first / tpp;
  if (NRM) {                                              // the prologue's first landing already reads the table
    w3_load_tab(Tb, a.aff, b0, a.B, tid);
    __syncthreads();
  }

  uintx4 wh[4], wl[4];
  w3_load_w<NS, false>(wh, wl, role ? J.W2 : J.W1, role ? J.ldw2 : J.ldw1, ch, g);
  const float bv = J.b1 ? J.b1[ch] : 0.f;
  const float* Xsrc = role ? J.X2 : J.X;
  const long xbs = role ? J.x2_bs : J.x_bs;

  const int u = tid & 511;
  const unsigned o0 = (unsigned)(4 * (u >> 4)) * P + 4 * (u & 15);
  const int oc = 4 * (u & 15);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  const W3Addr ad = w3_addr(ci, g);
  const int nloc = tl.nloc;

  W3Pre pre;
  pre.m = 0.f;
  auto issue = [&](int i) {   // unconditional, clamped (see w3_mlp_kernel)
    const int t = tl.first + (i < nloc ? i : nloc - 1) * tl.stride;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const float* Xt = Xsrc + (size_t)b * xbs + p0 + ((p0 + oc < P) ? o0 : 0u);
    pre.a = *reinterpret_cast<const float4*>(Xt);
    pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
    pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
    pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
    if (NRM) pre.m = a.maskstat[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
  };
  auto land = [&](int i) {
    const int t = tl.first + (i < nloc ? i : nloc - 1) * tl.stride;
    const int p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    const bool in = p0 + oc < P;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xa = in ? pre.a : z, xb = in ? pre.b : z, xc = in ? pre.c : z, xd = in ? pre.d : z;
    if (NRM && role == 1 && !(a.dbg & 2)) {   // the skip input's instance norm: channel 4 cg + v of graph b
      const float* T = Tb + (t / tpp - b0) * 256 + 4 * (u >> 4);
      const float4 sc = *reinterpret_cast<const float4*>(T), sh = *reinterpret_cast<const float4*>(T + 128);
      w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
    }
    w3_tile_store4<NS>(xa, xb, xc, xd, Xs, u);
    if (NRM && tid < 64) Ms[(i & 1) * 64 + tid] = (i < nloc && p0 + lane < P) ? pre.m : 0.f;
  };
  issue(0);
  land(0);
  issue(1);
  __syncthreads();
  for (int i = 0; i < nloc; ++i) {
    const int t = tl.first + i * tl.stride;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WS_TP);
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    w3_mma4<NS, false>(acc, wh, wl, Xs, ad);
    if (role == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Ps[j * 64] = ws_f4(acc[j]);
    }
    __syncthreads();
    land(i + 1);          // requested one tile ago
    issue(i + 2);
    if (role == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 q = Ps[j * 64];
        acc[j][0] += q.x; acc[j][1] += q.y; acc[j][2] += q.z; acc[j][3] += q.w;
      }
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
      if (NRM && !(a.dbg & 1)) {
        // statistics of the result over the tile's unmasked positions: the lane holds 16 positions of channel ch, the other 48
        // sit in the lanes that share ci.  One pass about a pivot that lies inside the data (w3_pivot), so that S2 - S1^2 / n loses
        // no digits to a large mean; three sums, two shuffle stages.  Selects, not products: a masked position may hold anything.
        const float* Mb = Ms + (i & 1) * 64;
        const float piv = w3_pivot(acc, Mb, lane, ci, bv);
        float s1 = 0.f, s2 = 0.f, n = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 mk = *reinterpret_cast<const float4*>(Mb + m0 + 16 * j);
          const float d0 = mk.x > 0.f ? acc[j][0] + bv - piv : 0.f, d1 = mk.y > 0.f ? acc[j][1] + bv - piv : 0.f;
          const float d2 = mk.z > 0.f ? acc[j][2] + bv - piv : 0.f, d3 = mk.w > 0.f ? acc[j][3] + bv - piv : 0.f;
          s1 += (d0 + d1) + (d2 + d3);
          s2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          n += (mk.x > 0.f ? 1.f : 0.f) + (mk.y > 0.f ? 1.f : 0.f) + (mk.z > 0.f ? 1.f : 0.f) + (mk.w > 0.f ? 1.f : 0.f);
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
          n += __shfl_xor(n, o, 64);
        }
        const float dm = s1 * __builtin_amdgcn_rcpf(fmaxf(n, 1.f));   // mean - pivot
        const float mean = piv + dm;
        const float q = fmaxf(s2 - s1 * dm, 0.f);
        if (g == 0) *reinterpret_cast<float2*>(a.stat + ((size_t)t * WS_H + ch) * 2) = make_float2(mean, q);
        if (tid == 0) a.cnt[t] = n;
      }
      // NRM: the raw result with masked positions zeroed (see w4_lin2_kernel's store)
      const float* Mo = NRM ? a.maskstat : a.maskp;
      ws_store_out(acc, J.Y + (size_t)b * J.y_bs + p0, r0, Mo ? Mo + (size_t)b * P + p0 : nullptr, m0, P - p0, bv, a.act != 0,
                   J.accumulate != 0);
    }
    __syncthreads();
  }
}

template <int NS>
inline int w3_lin2(const WsArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(NS > 1 ? 4 : 2) * W3_IMG + (size_t)WS_H * WS_TP * sizeof(float) + (2 * 64 + W3_TAB * 256) * sizeof(float);
  const bool nrm = a.stat != nullptr;
  if (nrm && (!a.cnt || !a.maskstat || !a.aff || a.njobs != 1)) return fail(DPPM_E_INVALID, "w3_lin2: statistics need aff, cnt, maskstat and a single job");
  auto kern = nrm ? w3_lin2_kernel<NS, true> : w3_lin2_kernel<NS, false>;
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds), "w3_lin2"));
  const int tpp = (a.P + WS_TP - 1) / WS_TP;
  const long ntiles = (long)a.B * tpp;
  const long slots = w3_slots(a.njobs, ntiles);
  if (nrm && !w3_contig_ok(ntiles, tpp, (int)(8 * slots))) return fail(DPPM_E_UNSUPPORTED, "w3_lin2: normalisation tables of a workgroup's tile run exceed LDS");
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.njobs * slots)), dim3(W3M_THREADS), lds, st, a);
  return check_launch("w3_lin2");
}

template <int NS> inline size_t w3_lds() { return (size_t)(NS > 1 ? 4 : 2) * W3_IMG; }   // two stages of hi (+ lo) images

template <int NS>
inline int w3_lin(const WsArgs& a, int nsrc, bool trans, hipStream_t st) {
  DPPM_TRY(ws_check(a, "w3_lin"));
  if (a.side && (!a.wside || !ws_al16(a.side) || a.side_bs % 4)) return fail(DPPM_E_INVALID, "w3_lin: bad side input");
  if (nsrc == 2)
    for (int i = 0; i < a.njobs; ++i)
      if (!a.job[i].X2 || !a.job[i].W2 || !ws_al16(a.job[i].X2) || a.job[i].x2_bs % 4) return fail(DPPM_E_INVALID, "w3_lin: job %d lacks its second source", i);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (nsrc == 1) rc = trans ? ws_launch_kernel(w3_lin_kernel<NS, 1, true>, w3_lds<NS>(), a, st, "w3_lin<1,T>")
                            : ws_launch_kernel(w3_lin_kernel<NS, 1, false>, w3_lds<NS>(), a, st, "w3_lin<1,N>");
  else if (!trans && (NS > 1 || a.stat)) rc = w3_lin2<NS>(a, st);   // the split form of two matrices does not fit one wave's registers; the fused-norm form lives there
  else rc = trans ? ws_launch_kernel(w3_lin_kernel<NS, 2, true>, w3_lds<NS>(), a, st, "w3_lin<2,T>")
                  : ws_launch_kernel(w3_lin_kernel<NS, 2, false>, w3_lds<NS>(), a, st, "w3_lin<2,N>");
  if (prof) prof_after(st, nsrc == 2 ? PROF_WS_SUM2 : PROF_WS_COLS, prof_arith_of_pieces(NS), 2.0 * nsrc * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, nsrc, false, false));
  return rc;
}

}  // namespace dppm
