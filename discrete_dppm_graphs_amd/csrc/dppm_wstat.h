// Weight-stationary kernels for the H = 128 PPGN linears (the bulk of the FLOPs: per layer 12 N^2 H^2 of
// 12 N^2 H^2 + 2 H N^3).  In channel-first layout a Linear is Y (128 x P) = W (128 x 128) * X (128 x P) with P = N*ld
// positions per graph: a tall-skinny problem whose only large dimension is P.  So the weights stay put and the
// positions stream through:
//   - a persistent workgroup (8 wave64, one per CU, 256 of them) walks 64-position tiles of all graphs;
//   - each wave keeps its 32 x 128 weight slice(s) in 64 VGPRs per matrix for the whole launch (loaded once);
//   - the X tile (128 channels x 64 positions, 32 KB) is staged through LDS, double buffered, the next tile's
//     global loads being issued before the current tile's MFMAs; all waves read it as the MFMA B operand with
//     conflict-free ds_read_b32;
//   - v_mfma_f32_32x32x2_f32 does the arithmetic (exact f32).  The k order inside a step is chosen as
//     k = ks + 64 * (lane >> 5) so that a lane's 64 weight values are one contiguous run of its row.
// Modes
//   MLP2 : waves 0-3 compute m1, waves 4-7 compute m2 (model/ppgn.py:58-59): hid = LeakyReLU(Wa X + ba) goes to LDS
//          (never to HBM in inference), then out = (Wb hid + bb) * mask.  One read of X, two writes.
//   SUM2 : Y = Wp * Xp + Wq * Xq + b, the skip-concat Linear m4 (model/ppgn.py:66-69) without materialising the cat:
//          every wave holds both 32 x 128 weight slices and runs one 256-deep chain; the wave groups split the columns.
//   COLS : Y = epilogue(W X + b) for one source, the two wave groups splitting a 128-position tile (after_conv).
#pragma once
#include <hip/hip_runtime.h>
#include "dppm_gemm.h"

namespace dppm {

constexpr int WS_H = 128;        // channels handled by these kernels
constexpr int WS_BP = 64;        // positions per tile and wave group
constexpr int WS_THREADS = 512;

enum WsMode { WS_MLP2 = 0, WS_SUM2 = 1, WS_COLS = 2 };

struct WsArgs {
  // sources: MLP2 uses X0 only; SUM2 uses X0 (with W0) and X1 (with W1); COLS uses X0
  const float* X0; long x0_bs;
  const float* X1; long x1_bs;
  // weights, torch [out][in] layout, row pitch ldw
  const float* W0; const float* W1;     // MLP2: first layers of m1 / m2;   SUM2: the two column blocks of m4;  COLS: W0
  const float* W2; const float* W3;     // MLP2: second layers of m1 / m2
  int ldw01, ldw23;
  const float* b0; const float* b1; const float* b2; const float* b3;   // biases matching W0..W3 (SUM2: b0 only)
  const float* maskp;                   // [B][P] or null (MLP2: applied to both outputs)
  int act;                              // COLS: LeakyReLU on the output
  float* Y0; float* Y1; long y_bs;      // MLP2: o1, o2;  SUM2 / COLS: Y0
  float* H0; float* H1;                 // MLP2 training: hidden activations saved for backward (else null)
  int B, P;
};

__device__ __forceinline__ int ws_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// one wave's 32 x 128 slice of W (rows row0..row0+31) into 64 registers: w[ks] = W[row0 + (lane&31)][64*(lane>>5) + ks]
__device__ __forceinline__ void ws_load_weights(float (&w)[64], const float* __restrict__ W, int ldw, int row0, int lane) {
  const float* p = W + (size_t)(row0 + (lane & 31)) * ldw + 64 * (lane >> 5);
  if ((ldw & 3) == 0 && (((uintptr_t)W) & 15) == 0) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float4 q = reinterpret_cast<const float4*>(p)[v];
      w[4 * v + 0] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = p[k];
  }
}

// acc (32 rows x 64 cols, two MFMA tiles) += Wslice * T, T an LDS tile [128][WS_BP]
__device__ __forceinline__ void ws_mma(floatx16& a0, floatx16& a1, const float (&w)[64], const float* __restrict__ T, int lane) {
  const float* t = T + (64 * (lane >> 5)) * WS_BP + (lane & 31);
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) {
    const float b0 = t[ks * WS_BP], b1 = t[ks * WS_BP + 32];
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ks], b0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ks], b1, a1, 0, 0, 0);
  }
}

// single-tile form: acc (32 x 32) += Wslice * T[:, col0 .. col0+31]  (T already offset to col0)
__device__ __forceinline__ void ws_mma1(floatx16& a0, const float (&w)[64], const float* __restrict__ T, int lane) {
  const float* t = T + (64 * (lane >> 5)) * WS_BP + (lane & 31);
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ks], t[ks * WS_BP], a0, 0, 0, 0);
}

// global tile (128 rows x 64 positions at p0) -> registers (4 float4 per thread of the 512) / registers -> LDS
__device__ __forceinline__ void ws_tile_load(float4 (&r)[4], const float* __restrict__ X, int P, int p0, int tid) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int idx = tid + v * WS_THREADS;
    const int k = idx >> 4, c = (idx & 15) * 4;
    r[v] = (p0 + c < P) ? *reinterpret_cast<const float4*>(X + (size_t)k * P + p0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ws_tile_store(const float4 (&r)[4], float* __restrict__ T, int tid) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int idx = tid + v * WS_THREADS;
    *reinterpret_cast<float4*>(T + (idx >> 4) * WS_BP + (idx & 15) * 4) = r[v];
  }
}

template <int MODE, bool SAVE_HID>
__global__ __launch_bounds__(WS_THREADS, 2) void wstat_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TILE = WS_H * WS_BP;                 // floats in one staged tile (32 KB)
  // LDS carve:  MLP2: X[2 stages], Hid[2 groups] ; SUM2: [2 stages][Xp, Xq] ; COLS: [2 stages][2 column halves] ; biases
  float* Xs = smem;
  float* Hs = smem + 2 * TILE;                       // MLP2 only
  float* Bs = smem + 4 * TILE;                       // biases [4][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, rg = wave & 3;            // wave group (0/1), row group (32 output rows each)
  const int P = a.P;
  const int tiles_per_plane = (P + (MODE == WS_COLS ? 2 * WS_BP : WS_BP) - 1) / (MODE == WS_COLS ? 2 * WS_BP : WS_BP);
  const int ntiles = a.B * tiles_per_plane;

  // ---- one-time: weights to registers, biases to LDS
  float wA[64], wB[64];
  if (MODE == WS_MLP2) {
    ws_load_weights(wA, g == 0 ? a.W0 : a.W1, a.ldw01, 32 * rg, lane);
    ws_load_weights(wB, g == 0 ? a.W2 : a.W3, a.ldw23, 32 * rg, lane);
  } else if (MODE == WS_SUM2) {
    ws_load_weights(wA, a.W0, a.ldw01, 32 * rg, lane);
    ws_load_weights(wB, a.W1, a.ldw01, 32 * rg, lane);
  } else {
    ws_load_weights(wA, a.W0, a.ldw01, 32 * rg, lane);
  }
  if (tid < WS_H) {
    Bs[tid] = a.b0 ? a.b0[tid] : 0.f;
    Bs[WS_H + tid] = (MODE == WS_MLP2 && a.b1) ? a.b1[tid] : 0.f;
    Bs[2 * WS_H + tid] = (MODE == WS_MLP2 && a.b2) ? a.b2[tid] : 0.f;
    Bs[3 * WS_H + tid] = (MODE == WS_MLP2 && a.b3) ? a.b3[tid] : 0.f;
  }

  auto tile_src = [&](int t, int which, const float*& X, int& p0) {   // which: 0 -> X0, 1 -> X1 (SUM2) / second half (COLS)
    const int b = t / tiles_per_plane, tp = t % tiles_per_plane;
    if (MODE == WS_COLS) { X = a.X0 + (size_t)b * a.x0_bs; p0 = tp * 2 * WS_BP + which * WS_BP; }
    else if (which == 0) { X = a.X0 + (size_t)b * a.x0_bs; p0 = tp * WS_BP; }
    else { X = a.X1 + (size_t)b * a.x1_bs; p0 = tp * WS_BP; }
  };
  constexpr int NSRC = (MODE == WS_MLP2) ? 1 : 2;    // staged tiles per step

  float4 pre[NSRC][4];
  int t = blockIdx.x;
  if (t < ntiles) {
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float* X; int p0;
      tile_src(t, s, X, p0);
      ws_tile_load(pre[s], X, P, p0, tid);
      ws_tile_store(pre[s], Xs + s * TILE, tid);       // stage 0
    }
  }
  __syncthreads();

  int cur = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
#pragma unroll
      for (int s = 0; s < NSRC; ++s) {
        const float* X; int p0;
        tile_src(tn, s, X, p0);
        ws_tile_load(pre[s], X, P, p0, tid);          // in flight during the MFMAs below
      }
    }
    const int b = t / tiles_per_plane, tp = t % tiles_per_plane;
    const float* Xcur = Xs + cur * NSRC * TILE;
    const int c = lane & 31;

    floatx16 acc0, acc1;
    if (MODE == WS_MLP2) {
      const int p0 = tp * WS_BP;
      // ---- stage 1: hid = LeakyReLU(Wa X + ba)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = Bs[g * WS_H + 32 * rg + ws_row(r, lane)];
      ws_mma(acc0, acc1, wA, Xcur, lane);
      float* Hg = Hs + g * TILE;
      float* Hglob = SAVE_HID ? (g == 0 ? a.H0 : a.H1) + (size_t)b * a.y_bs : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rg + ws_row(r, lane);
        float h0 = acc0[r], h1 = acc1[r];
        h0 = h0 > 0.f ? h0 : kSlope * h0;
        h1 = h1 > 0.f ? h1 : kSlope * h1;
        Hg[row * WS_BP + c] = h0;
        Hg[row * WS_BP + c + 32] = h1;
        if (SAVE_HID) {
          if (p0 + c < P) Hglob[(size_t)row * P + p0 + c] = h0;
          if (p0 + c + 32 < P) Hglob[(size_t)row * P + p0 + c + 32] = h1;
        }
      }
      __syncthreads();
      // ---- stage 2: out = (Wb hid + bb) * mask
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = Bs[(2 + g) * WS_H + 32 * rg + ws_row(r, lane)];
      ws_mma(acc0, acc1, wB, Hg, lane);
      float m0 = 1.f, m1 = 1.f;
      if (a.maskp) {
        m0 = (p0 + c < P) ? a.maskp[(size_t)b * P + p0 + c] : 0.f;
        m1 = (p0 + c + 32 < P) ? a.maskp[(size_t)b * P + p0 + c + 32] : 0.f;
      }
      float* Y = (g == 0 ? a.Y0 : a.Y1) + (size_t)b * a.y_bs;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rg + ws_row(r, lane);
        if (p0 + c < P) Y[(size_t)row * P + p0 + c] = acc0[r] * m0;
        if (p0 + c + 32 < P) Y[(size_t)row * P + p0 + c + 32] = acc1[r] * m1;
      }
    } else if (MODE == WS_SUM2) {
      const int p0 = tp * WS_BP + 32 * g;            // this wave group's 32 columns of the 64-position tile
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = Bs[32 * rg + ws_row(r, lane)];
      ws_mma1(acc0, wA, Xcur + 32 * g, lane);
      ws_mma1(acc0, wB, Xcur + TILE + 32 * g, lane);
      float* Y = a.Y0 + (size_t)b * a.y_bs;
      if (p0 + c < P) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[(size_t)(32 * rg + ws_row(r, lane)) * P + p0 + c] = acc0[r];
      }
    } else {   // WS_COLS
      const int p0 = tp * 2 * WS_BP + g * WS_BP;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = Bs[32 * rg + ws_row(r, lane)];
      ws_mma(acc0, acc1, wA, Xcur + g * TILE, lane);
      float m0 = 1.f, m1 = 1.f;
      if (a.maskp) {
        m0 = (p0 + c < P) ? a.maskp[(size_t)b * P + p0 + c] : 0.f;
        m1 = (p0 + c + 32 < P) ? a.maskp[(size_t)b * P + p0 + c + 32] : 0.f;
      }
      float* Y = a.Y0 + (size_t)b * a.y_bs;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rg + ws_row(r, lane);
        float v0 = acc0[r], v1 = acc1[r];
        if (a.act) { v0 = v0 > 0.f ? v0 : kSlope * v0; v1 = v1 > 0.f ? v1 : kSlope * v1; }
        if (p0 + c < P) Y[(size_t)row * P + p0 + c] = v0 * m0;
        if (p0 + c + 32 < P) Y[(size_t)row * P + p0 + c + 32] = v1 * m1;
      }
    }
    // ---- stage the prefetched tile(s) for the next step
    if (tn < ntiles) {
#pragma unroll
      for (int s = 0; s < NSRC; ++s) ws_tile_store(pre[s], Xs + ((cur ^ 1) * NSRC + s) * TILE, tid);
    }
    __syncthreads();
    cur ^= 1;
  }
}

inline size_t ws_lds_bytes(int mode) {
  const size_t tile = (size_t)WS_H * WS_BP * sizeof(float);
  (void)mode;
  return 4 * tile + 4 * WS_H * sizeof(float);
}

template <int MODE, bool SAVE_HID>
inline int ws_launch_one(const WsArgs& a, hipStream_t st) {
  static bool attr_set = false;
  const size_t lds = ws_lds_bytes(MODE);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wstat_kernel<MODE, SAVE_HID>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return fail(DPPM_E_LAUNCH, "wstat: cannot raise the dynamic LDS limit to %zu bytes", lds);
    attr_set = true;
  }
  const int per = (MODE == WS_COLS) ? 2 * WS_BP : WS_BP;
  const long ntiles = (long)a.B * ((a.P + per - 1) / per);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(wstat_kernel<MODE, SAVE_HID>), dim3(grid), dim3(WS_THREADS), lds, st, a);
  return check_launch("wstat");
}

inline double ws_flops(int mode, const WsArgs& a) {
  const double unit = 2.0 * WS_H * WS_H * (double)a.P * a.B;
  return mode == WS_MLP2 ? 4 * unit : (mode == WS_SUM2 ? 2 * unit : unit);
}

inline int ws_launch(int mode, bool save_hid, const WsArgs& a, hipStream_t st) {
  if (a.P % 4 != 0) return fail(DPPM_E_INVALID, "wstat: P=%d must be a multiple of 4", a.P);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (mode == WS_MLP2) rc = save_hid ? ws_launch_one<WS_MLP2, true>(a, st) : ws_launch_one<WS_MLP2, false>(a, st);
  else if (mode == WS_SUM2) rc = ws_launch_one<WS_SUM2, false>(a, st);
  else rc = ws_launch_one<WS_COLS, false>(a, st);
  if (prof) prof_after(st, mode == WS_MLP2 ? PROF_WS_MLP2 : (mode == WS_SUM2 ? PROF_WS_SUM2 : PROF_WS_COLS), ws_flops(mode, a));
  return rc;
}

}  // namespace dppm
