// Per-channel N x N contraction C[g] = op(A[g]) * op(B[g]) (model/ppgn.py:62 `out1 @ out2` and its two backward forms)
// for planes with N <= 256, on v_mfma_f32_16x16x4_f32 (exact f32).
//
// Shape regime: thousands of independent small square products (B*H planes, N = 20 ... 200), K = N.  A generic
// 128 x 128 tiling wastes 39 % of the matrix work on padding at N = 200 and a third of its time in per-tile prologues.
// Here one workgroup owns one whole plane: wave w computes the 16-row strip w of C across all NT = ceil(N/16) column
// tiles (NT accumulators of 4 registers), so padding is at most 15 rows/columns (N = 200: 208, 92 % useful), A and B are
// each read from HBM exactly once and C written once.  K advances 16 at a time through double-buffered LDS panels
// As[k][m] / Bs[k][n]; the next panels are in flight from HBM while the MFMAs of the current ones run.  Row pitch of the
// panels is 16 mod 32 floats so that the four 16-lane groups of a fragment read hit disjoint banks.  Several planes
// (workgroups) share a CU, which evens out the ceil(N/16) waves over the 4 SIMDs.
#pragma once
#include <hip/hip_runtime.h>
#include "dppm_gemm.h"
#include <cstdlib>

namespace dppm {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct PlaneArgs {
  const float* A; const float* B; float* C;
  int n, ld;               // planes are [n][ld]; K = n
  const float* scale;      // [batch / div] or null
  const float* mask;       // [batch / div][n][ld] or null
  int div;
};

template <int NT>
struct PlaneCfg {
  static constexpr int BX = NT * 16;
  static constexpr int LDT = (BX % 32 == 16) ? BX : BX + 16;     // pitch = 16 (mod 32): conflict-free fragment reads
  static constexpr int PANEL = BK * LDT;                          // floats per staged panel
  static constexpr int NVEC = BX * BK / 4;                        // float4 per panel
};

// operand panel: global -> registers.  KMAJOR: element (x,k) at base[k*ld + x]; else base[x*ld + k].  X = n for both.
template <int NT, bool KMAJOR, int NVPT>
__device__ __forceinline__ void plane_load(float4* r, const float* __restrict__ base, int ld, int n, int k0, int tid, int nthreads) {
  using Cfg = PlaneCfg<NT>;
#pragma unroll
  for (int v = 0; v < NVPT; ++v) {
    const int idx = tid + v * nthreads;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < Cfg::NVEC) {
      if (KMAJOR) {
        const int k = idx / (Cfg::BX / 4), x = (idx % (Cfg::BX / 4)) * 4;
        const int gk = k0 + k;
        if (gk < n && x < n) {
          const float* p = base + (size_t)gk * ld + x;
          val = *reinterpret_cast<const float4*>(p);     // ld % 4 == 0: pad columns (< ld) are readable; masked at store time
        }
      } else {
        const int x = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
        const int gk = k0 + k;
        if (x < n && gk < n) {
          const float* p = base + (size_t)x * ld + gk;
          val = *reinterpret_cast<const float4*>(p);
        }
      }
    }
    r[v] = val;
  }
}

template <int NT, bool KMAJOR, int NVPT>
__device__ __forceinline__ void plane_store(const float4* r, float* __restrict__ T, int n, int k0, int tid, int nthreads) {
  using Cfg = PlaneCfg<NT>;
#pragma unroll
  for (int v = 0; v < NVPT; ++v) {
    const int idx = tid + v * nthreads;
    if (idx < Cfg::NVEC) {
      float4 val = r[v];
      if (KMAJOR) {
        const int k = idx / (Cfg::BX / 4), x = (idx % (Cfg::BX / 4)) * 4;
        const int lim = n - x;
        if (lim < 2) val.y = 0.f;
        if (lim < 3) val.z = 0.f;
        if (lim < 4) val.w = 0.f;
        *reinterpret_cast<float4*>(&T[k * Cfg::LDT + x]) = val;
      } else {
        const int x = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
        const int lim = n - (k0 + k);
        T[(k + 0) * Cfg::LDT + x] = val.x;
        T[(k + 1) * Cfg::LDT + x] = lim < 2 ? 0.f : val.y;
        T[(k + 2) * Cfg::LDT + x] = lim < 3 ? 0.f : val.z;
        T[(k + 3) * Cfg::LDT + x] = lim < 4 ? 0.f : val.w;
      }
    }
  }
}

// One finished 16 x 16 tile, held TRANSPOSED (the MFMA was issued with its operands swapped: D = B_frag^T-by-A_frag, so
// the lane's four registers are four CONSECUTIVE COLUMNS of one row): row = row0 + (lane & 15), cols = col0 + 4 (lane >> 4) ..
// One 16-byte store per lane and tile instead of four 4-byte ones (the epilogue of the 4-byte form was store-issue bound:
// a third of the kernel at N = 200).  Columns in [n, ld) are written as zeros; ld % 4 == 0 keeps the vector inside the row.
__device__ __forceinline__ void plane_store_tile(const floatx4& acc, float* __restrict__ C, const float* __restrict__ mask, float scale,
                                                 int row0, int col0, int n, int ld, int lane) {
  const int row = row0 + (lane & 15), col = col0 + 4 * (lane >> 4);
  if (row < n && col < ld) {
    float4 v = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
    if (mask) {
      const float4 m = *reinterpret_cast<const float4*>(mask + (size_t)row * ld + col);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
    const int lim = n - col;
    if (lim < 1) v.x = 0.f;
    if (lim < 2) v.y = 0.f;
    if (lim < 3) v.z = 0.f;
    if (lim < 4) v.w = 0.f;
    *reinterpret_cast<float4*>(C + (size_t)row * ld + col) = v;
  }
}

// blockDim.x = 64 * ceil(n/16) (one wave per 16-row strip), grid.x = number of planes
// waves per SIMD to aim for: as many whole workgroups of NT waves as the CU's 32 wave slots hold (register budget permitting)
#ifndef DPPM_PLANE_FORCE_OCC
#define DPPM_PLANE_FORCE_OCC 0
#endif
constexpr int plane_waves_per_simd(int nt) { return (!DPPM_PLANE_FORCE_OCC || nt >= 14) ? (nt + 3) / 4 : ((32 / nt) * nt + 3) / 4; }

template <int NT, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void plane_mm_body(const PlaneArgs& p, const int g, float* __restrict__ lds) {
  using Cfg = PlaneCfg<NT>;
  constexpr int NVPT = 1;   // blockDim.x == 64 * NT threads move the NT*16 x 16 panel as one float4 each
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)p.n * p.ld;
  const float* A = p.A + g * plane;
  const float* B = p.B + g * plane;
  const int n = p.n, ld = p.ld;

  floatx4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};

  float4 ra[NVPT], rb[NVPT];
  plane_load<NT, A_KMAJOR, NVPT>(ra, A, ld, n, 0, tid, nthreads);
  plane_load<NT, B_KMAJOR, NVPT>(rb, B, ld, n, 0, tid, nthreads);
  plane_store<NT, A_KMAJOR, NVPT>(ra, lds, n, 0, tid, nthreads);
  plane_store<NT, B_KMAJOR, NVPT>(rb, lds + Cfg::PANEL, n, 0, tid, nthreads);
  __syncthreads();

  int stage = 0;
  for (int k0 = 0; k0 < n; k0 += BK) {
    const bool more = k0 + BK < n;
    if (more) {
      plane_load<NT, A_KMAJOR, NVPT>(ra, A, ld, n, k0 + BK, tid, nthreads);
      plane_load<NT, B_KMAJOR, NVPT>(rb, B, ld, n, k0 + BK, tid, nthreads);
    }
    const float* at = lds + stage * 2 * Cfg::PANEL + (lane >> 4) * Cfg::LDT + wave * 16 + (lane & 15);
    const float* bt = lds + stage * 2 * Cfg::PANEL + Cfg::PANEL + (lane >> 4) * Cfg::LDT + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const float af = at[ks * 4 * Cfg::LDT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bt[ks * 4 * Cfg::LDT + j * 16], af, acc[j], 0, 0, 0);   // operands swapped: transposed tile
    }
    if (more) {
      plane_store<NT, A_KMAJOR, NVPT>(ra, lds + (stage ^ 1) * 2 * Cfg::PANEL, n, k0 + BK, tid, nthreads);
      plane_store<NT, B_KMAJOR, NVPT>(rb, lds + (stage ^ 1) * 2 * Cfg::PANEL + Cfg::PANEL, n, k0 + BK, tid, nthreads);
    }
    __syncthreads();
    stage ^= 1;
  }

  const float scale = p.scale ? p.scale[g / p.div] : 1.f;
  const float* mask = p.mask ? p.mask + (size_t)(g / p.div) * plane : nullptr;
  float* C = p.C + g * plane;
#pragma unroll
  for (int j = 0; j < NT; ++j) plane_store_tile(acc[j], C, mask, scale, wave * 16, j * 16, n, ld, lane);
}
template <int NT, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NT * 64, plane_waves_per_simd(NT)) void plane_mm_kernel(const PlaneArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[4 * PlaneCfg<NT>::PANEL];   // [stage][A,B]
  plane_mm_body<NT, A_KMAJOR, B_KMAJOR>(p, blockIdx.x, lds);
}
// The two products of the contraction's backward in ONE launch (small networks, where a launch costs more than its work):
// workgroups [0, batch) form p1 = A * B^T (do1 = dmm o2^T), workgroups [batch, 2 batch) form p2 = A^T * B (do2 = o1^T dmm).
template <int NT>
__global__ __launch_bounds__(NT * 64, plane_waves_per_simd(NT)) void plane_mm_pair_kernel(const PlaneArgs p1, const PlaneArgs p2, int batch) {
  __shared__ __attribute__((aligned(16))) float lds[4 * PlaneCfg<NT>::PANEL];
  if ((int)blockIdx.x < batch) plane_mm_body<NT, false, false>(p1, blockIdx.x, lds);   // (block-uniform)
  else plane_mm_body<NT, true, true>(p2, blockIdx.x - batch, lds);
}
inline bool plane_mm_pair_ok(int n) { return n <= 32; }
inline int plane_mm_pair_launch(const PlaneArgs& p1, const PlaneArgs& p2, int batch, hipStream_t st) {
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  if (p1.n <= 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm_pair_kernel<1>), dim3(2 * batch), dim3(64), 0, st, p1, p2, batch);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm_pair_kernel<2>), dim3(2 * batch), dim3(128), 0, st, p1, p2, batch);
  if (prof) prof_after(st, PROF_PLANE_MM, AR_F32, 4.0 * p1.n * p1.n * (double)p1.n * batch, 6.0 * sizeof(float) * p1.n * (double)p1.ld * batch);
  return check_launch("plane_mm_pair");
}

template <int NT>
inline int plane_mm_launch_nt(const PlaneArgs& a, int batch, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  dim3 grid(batch), block(64 * NT);
  if (!a_kmajor && b_kmajor) hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm_kernel<NT, false, true>), grid, block, 0, st, a);
  else if (a_kmajor && b_kmajor) hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm_kernel<NT, true, true>), grid, block, 0, st, a);
  else if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm_kernel<NT, false, false>), grid, block, 0, st, a);
  else return fail(DPPM_E_UNSUPPORTED, "plane_mm: A^T * B^T is not instantiated");
  return check_launch("plane_mm");
}


// ---------------------------------------------------------------------------------------------------------
// 16-wave form for 9 <= ceil(N/16) <= 16 (N = 129 ... 256): the NT x NT grid of 16 x 16 output tiles is cut into 4 row
// groups x 4 column groups (sizes differ by at most one); wave (a, b) owns row group a and column group (a + b) mod 4.
// Waves with equal b share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically), so every SIMD receives one
// wave from each row group and each column group: a Latin square.  At N = 200 (NT = 13, groups 4,3,3,3) the SIMDs get
// 43/42/42/42 tiles instead of the 52/39/39/39 a strip-per-wave split would give, and each wave's A/B fragments are
// reused across its 3-4 tiles in the other direction (8 LDS reads per 16 MFMAs).
// ---------------------------------------------------------------------------------------------------------

template <int NT, int PBK>
struct Plane16Cfg {
  static constexpr int BX = NT * 16;
  static constexpr int LDT = (BX % 32 == 16) ? BX : BX + 16;
  static constexpr int PANEL = PBK * LDT;
  static constexpr int NVEC = BX * PBK / 4;
  static constexpr int NVPT = (NVEC + 1023) / 1024;
};

template <int NT, int PBK, bool KMAJOR>
__device__ __forceinline__ void p16_load(float4* r, const float* __restrict__ base, int ld, int n, int k0, int tid) {
  using Cfg = Plane16Cfg<NT, PBK>;
#pragma unroll
  for (int v = 0; v < Cfg::NVPT; ++v) {
    const int idx = tid + v * 1024;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < Cfg::NVEC) {
      int x, gk;
      const float* p;
      if (KMAJOR) { x = (idx % (Cfg::BX / 4)) * 4; gk = k0 + idx / (Cfg::BX / 4); p = base + (size_t)gk * ld + x; }
      else { x = idx / (PBK / 4); gk = k0 + (idx % (PBK / 4)) * 4; p = base + (size_t)x * ld + gk; }
      // ld % 4 == 0: the vector stays inside the row (pad columns are readable).  Nothing touches the loaded value
      // here, so the wait for it is deferred to p16_store, behind the MFMAs of the current panel.
      if (x < n && gk < n) val = *reinterpret_cast<const float4*>(p);
    }
    r[v] = val;
  }
}

// registers -> LDS panel; entries whose index along the contiguous direction is >= n are zeroed here
template <int NT, int PBK, bool KMAJOR>
__device__ __forceinline__ void p16_store(const float4* r, float* __restrict__ T, int n, int k0, int tid) {
  using Cfg = Plane16Cfg<NT, PBK>;
#pragma unroll
  for (int v = 0; v < Cfg::NVPT; ++v) {
    const int idx = tid + v * 1024;
    if (idx < Cfg::NVEC) {
      float4 val = r[v];
      if (KMAJOR) {
        const int x = (idx % (Cfg::BX / 4)) * 4;
        const int lim = n - x;
        if (lim < 2) val.y = 0.f;
        if (lim < 3) val.z = 0.f;
        if (lim < 4) val.w = 0.f;
        *reinterpret_cast<float4*>(&T[(idx / (Cfg::BX / 4)) * Cfg::LDT + x]) = val;
      } else {
        const int x = idx / (PBK / 4), k = (idx % (PBK / 4)) * 4;
        const int lim = n - (k0 + k);
        T[(k + 0) * Cfg::LDT + x] = val.x;
        T[(k + 1) * Cfg::LDT + x] = lim < 2 ? 0.f : val.y;
        T[(k + 2) * Cfg::LDT + x] = lim < 3 ? 0.f : val.z;
        T[(k + 3) * Cfg::LDT + x] = lim < 4 ? 0.f : val.w;
      }
    }
  }
}

// the whole K loop and epilogue for a wave that owns NR strips (from r0) x NC column tiles (from c0)
template <int NT, int PBK, int NR, int NC, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void p16_body(const PlaneArgs& p, float* lds, int g, int r0, int c0, int tid, int lane) {
  using Cfg = Plane16Cfg<NT, PBK>;
  const size_t plane = (size_t)p.n * p.ld;
  const float* A = p.A + g * plane;
  const float* B = p.B + g * plane;
  const int n = p.n, ld = p.ld;
  floatx4 acc[NR > 0 ? NR : 1][NC > 0 ? NC : 1];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  float4 ra[Cfg::NVPT], rb[Cfg::NVPT];
  p16_load<NT, PBK, A_KMAJOR>(ra, A, ld, n, 0, tid);
  p16_load<NT, PBK, B_KMAJOR>(rb, B, ld, n, 0, tid);
  p16_store<NT, PBK, A_KMAJOR>(ra, lds, n, 0, tid);
  p16_store<NT, PBK, B_KMAJOR>(rb, lds + Cfg::PANEL, n, 0, tid);
  __syncthreads();
  int stage = 0;
  for (int k0 = 0; k0 < n; k0 += PBK) {
    const bool more = k0 + PBK < n;
    if (more) {
      p16_load<NT, PBK, A_KMAJOR>(ra, A, ld, n, k0 + PBK, tid);
      p16_load<NT, PBK, B_KMAJOR>(rb, B, ld, n, k0 + PBK, tid);
    }
    if (NR > 0 && NC > 0) {
      const float* at = lds + stage * 2 * Cfg::PANEL + (lane >> 4) * Cfg::LDT + r0 * 16 + (lane & 15);
      const float* bt = lds + stage * 2 * Cfg::PANEL + Cfg::PANEL + (lane >> 4) * Cfg::LDT + c0 * 16 + (lane & 15);
      auto kstep = [&](int ks) {
        float af[NR > 0 ? NR : 1], bf[NC > 0 ? NC : 1];
#pragma unroll
        for (int i = 0; i < NR; ++i) af[i] = at[ks * 4 * Cfg::LDT + i * 16];
#pragma unroll
        for (int j = 0; j < NC; ++j) bf[j] = bt[ks * 4 * Cfg::LDT + j * 16];
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
          for (int j = 0; j < NC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);   // swapped: transposed tile
      };
#pragma unroll
      for (int ks = 0; ks < PBK / 4; ++ks) kstep(ks);
    }
    if (more) {
      p16_store<NT, PBK, A_KMAJOR>(ra, lds + (stage ^ 1) * 2 * Cfg::PANEL, n, k0 + PBK, tid);
      p16_store<NT, PBK, B_KMAJOR>(rb, lds + (stage ^ 1) * 2 * Cfg::PANEL + Cfg::PANEL, n, k0 + PBK, tid);
    }
    __syncthreads();
    stage ^= 1;
  }
  const float scale = p.scale ? p.scale[g / p.div] : 1.f;
  const float* mask = p.mask ? p.mask + (size_t)(g / p.div) * plane : nullptr;
  float* C = p.C + g * plane;
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) plane_store_tile(acc[i][j], C, mask, scale, (r0 + i) * 16, (c0 + j) * 16, n, ld, lane);
}

template <int NT, int PBK, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(1024) void plane_mm16_kernel(const PlaneArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds16[];
  constexpr int BASE = NT / 4, REM = NT % 4;       // group q has BASE + (q < REM) members and starts at q*BASE + min(q, REM)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = wave >> 2, b = wave & 3;
  const int cg = (a + b) & 3;
  const int r0 = a * BASE + (a < REM ? a : REM), c0 = cg * BASE + (cg < REM ? cg : REM);
  const bool rbig = a < REM, cbig = cg < REM;
  const int g = blockIdx.x;
  if (REM == 0) p16_body<NT, PBK, BASE, BASE, A_KMAJOR, B_KMAJOR>(p, lds16, g, r0, c0, tid, lane);
  else if (rbig && cbig) p16_body<NT, PBK, BASE + 1, BASE + 1, A_KMAJOR, B_KMAJOR>(p, lds16, g, r0, c0, tid, lane);
  else if (rbig) p16_body<NT, PBK, BASE + 1, BASE, A_KMAJOR, B_KMAJOR>(p, lds16, g, r0, c0, tid, lane);
  else if (cbig) p16_body<NT, PBK, BASE, BASE + 1, A_KMAJOR, B_KMAJOR>(p, lds16, g, r0, c0, tid, lane);
  else p16_body<NT, PBK, BASE, BASE, A_KMAJOR, B_KMAJOR>(p, lds16, g, r0, c0, tid, lane);
}

template <int NT, int PBK, bool AK, bool BKM>
inline int plane_mm16_launch_one(const PlaneArgs& a, int batch, hipStream_t st) {
  const size_t lds = 4 * (size_t)Plane16Cfg<NT, PBK>::PANEL * sizeof(float);
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(&plane_mm16_kernel<NT, PBK, AK, BKM>), (size_t)(lds), "plane_mm16"));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(plane_mm16_kernel<NT, PBK, AK, BKM>), dim3(batch), dim3(1024), lds, st, a);
  return check_launch("plane_mm16");
}

template <int NT, int PBK>
inline int plane_mm16_launch_pbk(const PlaneArgs& a, int batch, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  if (!a_kmajor && b_kmajor) return plane_mm16_launch_one<NT, PBK, false, true>(a, batch, st);
  if (a_kmajor && b_kmajor) return plane_mm16_launch_one<NT, PBK, true, true>(a, batch, st);
  if (!a_kmajor && !b_kmajor) return plane_mm16_launch_one<NT, PBK, false, false>(a, batch, st);
  return fail(DPPM_E_UNSUPPORTED, "plane_mm16: A^T * B^T is not instantiated");
}

// K panel depth: 32 halves the barriers, 16 wastes less when n is just past a multiple of 16 (n = 200: 208 vs 224)
template <int NT>
inline int plane_mm16_launch_nt(const PlaneArgs& a, int batch, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  if (round_up(a.n, 32) == round_up(a.n, 16)) return plane_mm16_launch_pbk<NT, 32>(a, batch, a_kmajor, b_kmajor, st);
  return plane_mm16_launch_pbk<NT, 16>(a, batch, a_kmajor, b_kmajor, st);
}

inline bool plane_mm_supported(int n) { return n <= 256; }

inline int plane_mm_launch(const PlaneArgs& a, int batch, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  const int nt = cdiv(a.n, 16);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  const bool strips_only = sw(SW_PLANE_STRIPS_ONLY) != 0;   // A/B switch for measurements
  switch (strips_only ? -nt : nt) {
#define DPPM_PLANE_CASE(NT_) case NT_: case -NT_: rc = plane_mm_launch_nt<NT_>(a, batch, a_kmajor, b_kmajor, st); break;
#define DPPM_PLANE16_CASE(NT_) case NT_: rc = plane_mm16_launch_nt<NT_>(a, batch, a_kmajor, b_kmajor, st); break; \
                               case -NT_: rc = plane_mm_launch_nt<NT_>(a, batch, a_kmajor, b_kmajor, st); break;
    DPPM_PLANE_CASE(1) DPPM_PLANE_CASE(2) DPPM_PLANE_CASE(3) DPPM_PLANE_CASE(4) DPPM_PLANE_CASE(5) DPPM_PLANE_CASE(6)
    DPPM_PLANE_CASE(7) DPPM_PLANE_CASE(8) DPPM_PLANE16_CASE(9) DPPM_PLANE16_CASE(10) DPPM_PLANE16_CASE(11) DPPM_PLANE16_CASE(12)
    DPPM_PLANE16_CASE(13) DPPM_PLANE16_CASE(14) DPPM_PLANE16_CASE(15) DPPM_PLANE16_CASE(16)
#undef DPPM_PLANE_CASE
#undef DPPM_PLANE16_CASE
    default: rc = fail(DPPM_E_UNSUPPORTED, "plane_mm: n=%d exceeds the strip kernel's range", a.n);
  }
  if (prof) prof_after(st, PROF_PLANE_MM, AR_F32, 2.0 * a.n * a.n * (double)a.n * batch, 3.0 * sizeof(float) * a.n * (double)a.ld * batch);
  return rc;
}

}  // namespace dppm
