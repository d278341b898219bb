// Weight gradients of the H = 128 linears: dW (128 x 128) = sum_{b,p} dY[b][:, p] X[b][:, p]^T, db = sum_{b,p} dY[b][:, p].
// The reduction runs over all B * P positions (1.3 M at N = 200, B = 32): a split-K problem with a tiny output.
//   - 512 persistent workgroups (two per CU) stride through 64-position tiles and keep their 128 x 128 partial product
//     in registers (8 waves x 32 accumulator registers) for the whole launch: one slab per workgroup, not per tile;
//   - both operands are staged in their NATURAL layout ([channel][position], a straight 16-byte copy): with a row pitch of
//     68 floats the MFMA fragment reads (16 channels x 4 consecutive positions per wave) touch 64 distinct banks, so the
//     transposition an operand with K contiguous usually needs never happens;
//   - v_mfma_f32_16x16x4_f32 with k = position: D[m][n] += dY[m][p] * X[n][p];
//   - slabs are folded by a two-stage reduction in a fixed order (bitwise reproducible run to run).
#pragma once
#include <hip/hip_runtime.h>
#include "dppm_gemm.h"
#include "dppm_wstat.h"
#include "dppm_wsplit.h"

namespace dppm {

constexpr int WG_TP = 64;              // positions per tile
constexpr int WG_PITCH = 68;           // LDS row pitch (floats)
constexpr int WG_THREADS = 512;
constexpr int WG_GRID = 512;           // slabs per launch
constexpr int WG_SLAB = WS_H * WS_H + WS_H;   // floats per slab: dW then db
constexpr int WG_ZC = 32;              // first-stage groups of the slab reduction

struct WgArgs {
  const float* dY; long dy_bs;         // [B][128][P]
  const float* X;  long x_bs;          // [B][128][P]
  float* slabs;                        // [gridDim.x][WG_SLAB]
  int B, P;
};

__device__ __forceinline__ void wg_tile_load(float4 (&r)[4], const float* __restrict__ Xt, unsigned P, unsigned o0, bool ok) {
#pragma unroll
  for (int v = 0; v < 4; ++v) r[v] = ok ? *reinterpret_cast<const float4*>(Xt + (size_t)(32 * v) * P + o0) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void wg_tile_store(const float4 (&r)[4], float* __restrict__ T, int tid) {
  float* q = T + (tid >> 4) * WG_PITCH + (tid & 15) * 4;
#pragma unroll
  for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(q + 32 * v * WG_PITCH) = r[v];
}

template <bool WITH_DB>
__global__ __launch_bounds__(WG_THREADS, 4) void ws_wgrad_kernel(const WgArgs a) {
  __shared__ __attribute__((aligned(16))) float Ys[WS_H * WG_PITCH];
  __shared__ __attribute__((aligned(16))) float Xs[WS_H * WG_PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;           // this wave: rows 64 wm .. +63 of dW, columns 32 wn .. +31
  const int P = a.P, tpp = (P + WG_TP - 1) / WG_TP, ntiles = a.B * tpp;
  const unsigned o0 = (unsigned)(tid >> 4) * P + 4 * (tid & 15);
  const int oc = 4 * (tid & 15);

  floatx4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float dbv[4] = {0.f, 0.f, 0.f, 0.f};               // row sums of dY for rows (tid >> 4) + 32 v

  float4 py[4], px[4];
  int t = blockIdx.x;
  if (t < ntiles) {
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * WG_TP);
    wg_tile_load(py, a.dY + (size_t)b * a.dy_bs + p0, P, o0, p0 + oc < P);
    wg_tile_load(px, a.X + (size_t)b * a.x_bs + p0, P, o0, p0 + oc < P);
    wg_tile_store(py, Ys, tid);
    wg_tile_store(px, Xs, tid);
    if (WITH_DB) {
#pragma unroll
      for (int v = 0; v < 4; ++v) dbv[v] += (py[v].x + py[v].y) + (py[v].z + py[v].w);
    }
  }
  __syncthreads();

  const float* ya = Ys + (64 * wm + ci) * WG_PITCH + g;    // A fragment: channel 64 wm + 16 i + ci, position 4 k + g
  const float* xb = Xs + (32 * wn + ci) * WG_PITCH + g;    // B fragment: channel 32 wn + 16 j + ci, position 4 k + g
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
      const int b = __builtin_amdgcn_readfirstlane(tn / tpp), p0 = __builtin_amdgcn_readfirstlane((tn % tpp) * WG_TP);
      wg_tile_load(py, a.dY + (size_t)b * a.dy_bs + p0, P, o0, p0 + oc < P);
      wg_tile_load(px, a.X + (size_t)b * a.x_bs + p0, P, o0, p0 + oc < P);
    }
#pragma unroll
    for (int k = 0; k < WG_TP / 4; ++k) {
      float af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = ya[16 * i * WG_PITCH + 4 * k];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = xb[16 * j * WG_PITCH + 4 * k];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (tn < ntiles) {
      wg_tile_store(py, Ys, tid);
      wg_tile_store(px, Xs, tid);
      if (WITH_DB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) dbv[v] += (py[v].x + py[v].y) + (py[v].z + py[v].w);
      }
    }
    __syncthreads();
  }

  // this workgroup's slab: D[m = 4 g + r][n = ci] of tile (i, j)
  float* slab = a.slabs + (size_t)blockIdx.x * WG_SLAB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(64 * wm + 16 * i + 4 * g + r) * WS_H + 32 * wn + 16 * j + ci] = acc[i][j][r];
  if (WITH_DB) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float s = dbv[v];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      if ((tid & 15) == 0) slab[WS_H * WS_H + (tid >> 4) + 32 * v] = s;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Three-piece split form of the same product on the bf16 matrix pipe: every float32 operand is written as
//   x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (3 x 8 significand bits = all 24)
// and a product is hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid on v_mfma_f32_16x16x32_bf16 with float32 accumulation:
// the dropped terms are <= 2^-24 of the product, i.e. float32-grade, at 6/16 of the f32 MFMA time, which this HBM-bound
// kernel does not feel.  (The two-piece form of dppm_wsplit.h, 2^-16 per product, is NOT good enough here: where the 1.3 M
// summed terms cancel, it landed 6x further from float64 than exact-f32 products; measured on test_backward_kinkfree_h128.)
// K is the position index, contiguous in both operands' natural layout, so a lane's MFMA fragment (8 consecutive positions
// of one channel) needs no transpose: a staged tile is an image [128 channels][32 positions] of bf16 per piece (64-byte
// rows, chunk swizzle c ^ ((-(r >> 2)) & 3): conflict-free ds_read_b128; 16 staging lanes write two whole rows).
// ---------------------------------------------------------------------------------------------------------
constexpr int W6_TP = 32;                      // positions per tile = one MFMA K step
constexpr int W6_IMG = WS_H * W6_TP * 2;       // bytes of one piece image (8 KB)

__device__ __forceinline__ unsigned wg6_off(unsigned row, unsigned c) { return row * 64 + ((c ^ ((0u - (row >> 2)) & 3u)) << 4); }

struct Wg6Pre { float4 a, b; };

__device__ __forceinline__ void wg6_issue(Wg6Pre& q, const float* __restrict__ Xt, unsigned P, unsigned o) {
  q.a = *reinterpret_cast<const float4*>(Xt + o);
  q.b = *reinterpret_cast<const float4*>(Xt + (size_t)64 * P + o);
}
__device__ __forceinline__ void wg6_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = w3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = w3_pack(r0, r1);
  l = w3_pack(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void wg6_put(char* __restrict__ img, unsigned off, const float4& v, bool in) {
  unsigned h01, m01, l01, h23, m23, l23;
  wg6_split2(in ? v.x : 0.f, in ? v.y : 0.f, h01, m01, l01);
  wg6_split2(in ? v.z : 0.f, in ? v.w : 0.f, h23, m23, l23);
  *reinterpret_cast<uint2*>(img + off) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(img + W6_IMG + off) = make_uint2(m01, m23);
  *reinterpret_cast<uint2*>(img + 2 * W6_IMG + off) = make_uint2(l01, l23);
}
// thread (row = tid >> 3 (+ 64), position quad q = tid & 7): 4 positions = 8 bytes, chunk q >> 1, half q & 1
__device__ __forceinline__ void wg6_land(const Wg6Pre& p, char* __restrict__ img, int tid, bool in) {
  const unsigned row = tid >> 3, q = tid & 7, hb = (q & 1) * 8, c = q >> 1;
  wg6_put(img, wg6_off(row, c) + hb, p.a, in);
  wg6_put(img, wg6_off(row + 64, c) + hb, p.b, in);
}

template <bool WITH_DB>
__global__ __launch_bounds__(WG_THREADS, 4) void ws6_wgrad_kernel(const WgArgs a) {
  __shared__ __attribute__((aligned(16))) char Ys[3 * W6_IMG];   // hi, mid, lo images of the dY tile
  __shared__ __attribute__((aligned(16))) char Xs[3 * W6_IMG];   // hi, mid, lo images of the X tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;           // this wave: rows 64 wm .. +63 of dW, columns 32 wn .. +31
  const int P = a.P, tpp = (P + W6_TP - 1) / W6_TP, ntiles = a.B * tpp;
  const unsigned o0 = (unsigned)(tid >> 3) * P + 4 * (tid & 7);
  const int oc = 4 * (tid & 7);
  const int nloc = ((int)ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1: the grid never exceeds the tile count

  floatx4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float dbv[2] = {0.f, 0.f};                         // row sums of dY for rows (tid >> 3) + 64 v

  Wg6Pre py, px;
  bool pin = false;
  // unconditional, clamped loads (counted waits stay exact, see dppm_wsplit.h); tiles past the end re-read the last one
  auto issue = [&](int i) {
    const int t = (int)blockIdx.x + (i < nloc ? i : nloc - 1) * (int)gridDim.x;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t % tpp) * W6_TP);
    pin = p0 + oc < P;
    const unsigned o = pin ? o0 : 0u;
    wg6_issue(py, a.dY + (size_t)b * a.dy_bs + p0, P, o);
    wg6_issue(px, a.X + (size_t)b * a.x_bs + p0, P, o);
  };
  auto land = [&]() {
    wg6_land(py, Ys, tid, pin);
    wg6_land(px, Xs, tid, pin);
    if (WITH_DB && pin) {
      dbv[0] += (py.a.x + py.a.y) + (py.a.z + py.a.w);
      dbv[1] += (py.b.x + py.b.y) + (py.b.z + py.b.w);
    }
  };
  issue(0);
  land();
  __syncthreads();
  for (int i = 0; i < nloc; ++i) {
    const bool more = i + 1 < nloc;
    issue(i + 1);
    uintx4 bh[2], bm[2], bl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned bo = wg6_off(32 * wn + 16 * j + ci, g);
      bh[j] = *reinterpret_cast<const uintx4*>(Xs + bo);
      bm[j] = *reinterpret_cast<const uintx4*>(Xs + W6_IMG + bo);
      bl[j] = *reinterpret_cast<const uintx4*>(Xs + 2 * W6_IMG + bo);
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const unsigned ao = wg6_off(64 * wm + 16 * ii + ci, g);
      const uintx4 ah = *reinterpret_cast<const uintx4*>(Ys + ao);
      const uintx4 am = *reinterpret_cast<const uintx4*>(Ys + W6_IMG + ao);
      const uintx4 al = *reinterpret_cast<const uintx4*>(Ys + 2 * W6_IMG + ao);
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // smallest terms first
        acc[ii][j] = w3_mfma(am, bm[j], acc[ii][j]);
        acc[ii][j] = w3_mfma(al, bh[j], acc[ii][j]);
        acc[ii][j] = w3_mfma(ah, bl[j], acc[ii][j]);
        acc[ii][j] = w3_mfma(am, bh[j], acc[ii][j]);
        acc[ii][j] = w3_mfma(ah, bm[j], acc[ii][j]);
        acc[ii][j] = w3_mfma(ah, bh[j], acc[ii][j]);
      }
    }
    __syncthreads();
    if (more) land();      // the clamped re-read after the last tile is dropped
    __syncthreads();
  }

  // this workgroup's slab: D[m = 4 g + r][n = ci] of tile (i, j)
  float* slab = a.slabs + (size_t)blockIdx.x * WG_SLAB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(64 * wm + 16 * i + 4 * g + r) * WS_H + 32 * wn + 16 * j + ci] = acc[i][j][r];
  if (WITH_DB) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      float s = dbv[v];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if ((tid & 7) == 0) slab[WS_H * WS_H + (tid >> 3) + 64 * v] = s;
    }
  }
}

// stage 1: part[zc][e] = sum of slabs zc * per .. (zc + 1) * per - 1 ; stage 2: out = sum_zc part[zc][e]   (fixed orders)
__global__ __launch_bounds__(256) void wg_reduce1_kernel(const float* __restrict__ slabs, int nz, int per, int ne, int stride,
                                                         float* __restrict__ part) {
  const int e = blockIdx.x * 256 + threadIdx.x, zc = blockIdx.y;
  if (e >= ne) return;
  const int z0 = zc * per, z1 = z0 + per < nz ? z0 + per : nz;
  float s = 0.f;
  for (int z = z0; z < z1; ++z) s += slabs[(size_t)z * stride + e];
  part[(size_t)zc * ne + e] = s;
}
__global__ __launch_bounds__(256) void wg_reduce2_kernel(const float* __restrict__ part, int nzc, int ne, float* __restrict__ dW, int ldo,
                                                         float* __restrict__ db) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  float s = 0.f;
  for (int z = 0; z < nzc; ++z) s += part[(size_t)z * ne + e];
  if (e < WS_H * WS_H) dW[(size_t)(e / WS_H) * ldo + (e % WS_H)] = s;
  else if (db) db[e - WS_H * WS_H] = s;
}

inline size_t ws_wgrad_scratch_floats() { return (size_t)WG_GRID * WG_SLAB + (size_t)WG_ZC * WG_SLAB; }

// dW (128 x 128, row pitch ldo) and optionally db from dY and X, both [B][128][P].  scratch: ws_wgrad_scratch_floats().
inline int ws_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int B, int P, float* dW, int ldo, float* db, float* scratch,
                    hipStream_t st, bool split = false) {
  if (!dY || !X || !dW || !scratch || B <= 0 || P <= 0 || P % 4 != 0) return fail(DPPM_E_INVALID, "ws_wgrad: bad arguments");
  if (!ws_al16(dY) || !ws_al16(X) || dy_bs % 4 != 0 || x_bs % 4 != 0) return fail(DPPM_E_INVALID, "ws_wgrad: planes must be 16-byte aligned");
  if ((long)WS_H * P >= (1L << 31)) return fail(DPPM_E_INVALID, "ws_wgrad: P=%d too large", P);
  const long ntiles = (long)B * ((P + (split ? W6_TP : WG_TP) - 1) / (split ? W6_TP : WG_TP));
  const int grid = (int)(ntiles < WG_GRID ? ntiles : WG_GRID);
  WgArgs a{dY, dy_bs, X, x_bs, scratch, B, P};
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  if (split && db) hipLaunchKernelGGL(ws6_wgrad_kernel<true>, dim3(grid), dim3(WG_THREADS), 0, st, a);
  else if (split) hipLaunchKernelGGL(ws6_wgrad_kernel<false>, dim3(grid), dim3(WG_THREADS), 0, st, a);
  else if (db) hipLaunchKernelGGL(ws_wgrad_kernel<true>, dim3(grid), dim3(WG_THREADS), 0, st, a);
  else hipLaunchKernelGGL(ws_wgrad_kernel<false>, dim3(grid), dim3(WG_THREADS), 0, st, a);
  if (prof) prof_after(st, PROF_WGRAD, split ? AR_F32X6 : AR_F32, 2.0 * WS_H * WS_H * (double)P * B, 2.0 * sizeof(float) * WS_H * (double)P * B);
  DPPM_TRY(check_launch("ws_wgrad"));
  const int ne = db ? WG_SLAB : WS_H * WS_H;
  const int per = (grid + WG_ZC - 1) / WG_ZC, nzc = (grid + per - 1) / per;
  float* part = scratch + (size_t)WG_GRID * WG_SLAB;
  hipLaunchKernelGGL(wg_reduce1_kernel, dim3((ne + 255) / 256, nzc), dim3(256), 0, st, scratch, grid, per, ne, WG_SLAB, part);
  DPPM_TRY(check_launch("wg_reduce1"));
  hipLaunchKernelGGL(wg_reduce2_kernel, dim3((ne + 255) / 256), dim3(256), 0, st, part, nzc, ne, dW, ldo, db);
  return check_launch("wg_reduce2");
}

}  // namespace dppm
