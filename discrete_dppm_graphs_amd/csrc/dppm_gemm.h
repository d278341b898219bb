// Batched float32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain) with
// the fused epilogues the PPGN needs.  One kernel family serves
//   - the 1x1-conv linears in channel-first layout:   C[b] (O x P) = W (O x I) * X[b] (I x P)      (model/ppgn.py:58-59,69)
//   - the per-channel adjacency contraction:          C[g] (N x N) = A[g] * B[g]                     (model/ppgn.py:62)
//   - their backward forms (A^T*B, A*B^T, W^T*dY) and the weight gradients dW = sum_b dY[b] * X[b]^T (split-K).
//
// Tiling: a 256-thread workgroup (4 wave64) owns a BM x BN output tile; each wave a TM x TN sub-tile built from
// 32x32 MFMA tiles held in accumulator registers.  The K loop advances BK = 16 at a time through two LDS stages:
// while the matrix cores consume stage s, the next A/B slabs are already in flight from HBM/L2 into registers and
// are written to stage s^1 after the MFMAs (one barrier per K step).  Operands may be stored "K-major" (rows
// indexed by k, contiguous along m or n: copied to LDS with 16-byte writes) or "K-minor" (rows indexed by m or n,
// contiguous along k: transposed on the way into LDS; the row pitch BX+2 keeps those 4-byte writes conflict
// free), so no transposed copy of any activation is ever materialised.  LDS reads are ds_read_b32 with lanes
// 0..31 on consecutive addresses (conflict free).
#pragma once
#include <hip/hip_runtime.h>
#include "dppm_common.h"

namespace dppm {

constexpr float kSlope = 0.01f;   // LeakyReLU negative slope, model/ppgn.py:7
constexpr int BK = 16;

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmSeg {
  const float* A; long a_bs; int lda;   // A operand: M x K
  const float* B; long b_bs; int ldb;   // B operand: K x N
  int K;
};

struct GemmArgs {
  GemmSeg seg[2];
  int nseg;
  int M, N;
  int n_store;             // columns written per row (>= N; columns in [N, n_store) get zeros), 0 = N
  int a_vec, b_vec;        // 16-byte vector loads allowed (base, ld and offsets 16-byte aligned)
  int nsplit, kchunk;      // split-K over segment 0 (nsplit > 1): blockIdx.z = g * nsplit + split
  float* C; long c_bs; int ldc;
  const float* bias;       // [M] or null
  const float* scale;      // [batch / scale_div] or null
  int scale_div;
  int act;                 // 1: LeakyReLU
  const float* mul;        // elementwise multiplier: mul[(g / mul_div) * mul_bs + m * mul_ld + n], or null
  long mul_bs; int mul_ld; int mul_div;
  const float* dact;       // same layout as C: multiply by LeakyReLU'(dact)
  long dact_bs;
  int accumulate;          // C += result
  int prof_class;          // ProfClass of this launch (measurement hooks only)
  float* rowsum;           // [gridDim.z][M] row sums of the A operand over this block's K range (K-minor A only), or null
};

template <int BX, bool KMAJOR>
struct TileCfg {
  static constexpr int LDT = KMAJOR ? BX : BX + 2;
  static constexpr int NVEC = BX * BK / 4;                       // float4 per tile
  static constexpr int PER_THREAD = (NVEC + 255) / 256;
  static constexpr int LDS_FLOATS = BK * LDT;
};

// global -> registers.  base points at the operand for this batch; x0 = first row/col of the tile in M or N,
// X = logical extent in that dimension, [k0, kend) the K range still to be covered by this slab.
template <int BX, bool KMAJOR>
__device__ __forceinline__ void tile_load(float4* r, const float* __restrict__ base,
                                          int ld, int x0, int X, int k0, int kend, bool vec, int tid) {
  using Cfg = TileCfg<BX, KMAJOR>;
#pragma unroll
  for (int v = 0; v < Cfg::PER_THREAD; ++v) {
    const int idx = tid + v * 256;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (Cfg::NVEC % 256 == 0 || idx < Cfg::NVEC) {
      if (KMAJOR) {
        const int k = idx / (BX / 4), x = (idx % (BX / 4)) * 4;
        const int gk = k0 + k, gx = x0 + x;
        if (gk < kend && gx < X) {
          const float* p = base + (size_t)gk * ld + gx;
          if (vec && gx + 3 < X) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gx + 1 < X) val.y = p[1];
            if (gx + 2 < X) val.z = p[2];
            if (gx + 3 < X) val.w = p[3];
          }
        }
      } else {
        const int x = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
        const int gx = x0 + x, gk = k0 + k;
        if (gx < X && gk < kend) {
          const float* p = base + (size_t)gx * ld + gk;
          if (vec && gk + 3 < kend) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gk + 1 < kend) val.y = p[1];
            if (gk + 2 < kend) val.z = p[2];
            if (gk + 3 < kend) val.w = p[3];
          }
        }
      }
    }
    r[v] = val;
  }
}

// registers -> LDS stage (layout T[k][x], pitch LDT)
template <int BX, bool KMAJOR>
__device__ __forceinline__ void tile_store(const float4* r, float* __restrict__ T, int tid) {
  using Cfg = TileCfg<BX, KMAJOR>;
#pragma unroll
  for (int v = 0; v < Cfg::PER_THREAD; ++v) {
    const int idx = tid + v * 256;
    if (Cfg::NVEC % 256 == 0 || idx < Cfg::NVEC) {
      if (KMAJOR) {
        const int k = idx / (BX / 4), x = (idx % (BX / 4)) * 4;
        *reinterpret_cast<float4*>(&T[k * Cfg::LDT + x]) = r[v];
      } else {
        const int x = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
        T[(k + 0) * Cfg::LDT + x] = r[v].x;
        T[(k + 1) * Cfg::LDT + x] = r[v].y;
        T[(k + 2) * Cfg::LDT + x] = r[v].z;
        T[(k + 3) * Cfg::LDT + x] = r[v].w;
      }
    }
  }
}

// TAG only separates otherwise identical instantiations into distinct kernel symbols (0 linear, 1 plane, 2 wgrad) so that
// rocprofv3 --stats reports them separately.
template <int BM, int BN, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, int TAG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM, TN = BN / WN;      // per-wave tile
  constexpr int FM = TM / 32, FN = TN / 32;      // 32x32 MFMA tiles per wave
  using ACfg = TileCfg<BM, A_KMAJOR>;
  using BCfg = TileCfg<BN, B_KMAJOR>;
  __shared__ __attribute__((aligned(16))) float lds[2 * (ACfg::LDS_FLOATS + BCfg::LDS_FLOATS)];
  constexpr int STAGE_FLOATS = ACfg::LDS_FLOATS + BCfg::LDS_FLOATS;   // stage s: A slab at s*STAGE, B slab behind it

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: tile_ok branches stay scalar
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
  const int g = z / p.nsplit, split = z % p.nsplit;

  const int nstore = p.n_store > p.N ? p.n_store : p.N;

  floatx16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // The MFMA chain is kept free of branches (a conditional MFMA makes hipcc shuttle the accumulators through VGPRs
  // around every instruction); waves whose tile lies outside the output multiply zero-filled slabs and skip the store.
  const bool wave_ok = (m0 + wm * TM < p.M) && (n0 + wn * TN < nstore);

  float rs[ACfg::PER_THREAD];
#pragma unroll
  for (int v = 0; v < ACfg::PER_THREAD; ++v) rs[v] = 0.f;

  float4 ra[ACfg::PER_THREAD], rb[BCfg::PER_THREAD];

  // flattened (segment, k) iteration space
  int s = 0;
  int k = (p.nsplit > 1) ? split * p.kchunk : 0;
  int kend = (p.nsplit > 1) ? min(p.seg[0].K, k + p.kchunk) : p.seg[0].K;
  const float* Ab = p.seg[0].A + (size_t)g * p.seg[0].a_bs;
  const float* Bb = p.seg[0].B + (size_t)g * p.seg[0].b_bs;
  int lda = p.seg[0].lda, ldb = p.seg[0].ldb;
  const bool avec = p.a_vec, bvec = p.b_vec;

  auto advance = [&]() -> bool {   // move (s,k) to the next slab; false when exhausted
    k += BK;
    if (k < kend) return true;
    if (p.nsplit > 1 || s + 1 >= p.nseg) return false;
    ++s;
    k = 0;
    kend = p.seg[1].K;
    Ab = p.seg[1].A + (size_t)g * p.seg[1].a_bs;
    Bb = p.seg[1].B + (size_t)g * p.seg[1].b_bs;
    lda = p.seg[1].lda;
    ldb = p.seg[1].ldb;
    return kend > 0;
  };

  bool have = k < kend;
  if (have) {
    tile_load<BM, A_KMAJOR>(ra, Ab, lda, m0, p.M, k, kend, avec, tid);
    tile_load<BN, B_KMAJOR>(rb, Bb, ldb, n0, p.N, k, kend, bvec, tid);
    if (p.rowsum && !A_KMAJOR) {
#pragma unroll
      for (int v = 0; v < ACfg::PER_THREAD; ++v) rs[v] += (ra[v].x + ra[v].y) + (ra[v].z + ra[v].w);
    }
    tile_store<BM, A_KMAJOR>(ra, lds, tid);
    tile_store<BN, B_KMAJOR>(rb, lds + ACfg::LDS_FLOATS, tid);
  }
  __syncthreads();

  int stage = 0;
  while (have) {
    const bool more = advance();
    if (more) {   // next slab: global loads in flight during the MFMAs below
      tile_load<BM, A_KMAJOR>(ra, Ab, lda, m0, p.M, k, kend, avec, tid);
      tile_load<BN, B_KMAJOR>(rb, Bb, ldb, n0, p.N, k, kend, bvec, tid);
    }
    const float* at = lds + stage * STAGE_FLOATS + wm * TM + (lane & 31);
    const float* bt = lds + stage * STAGE_FLOATS + ACfg::LDS_FLOATS + wn * TN + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int kk = 2 * ks + (lane >> 5);
      float af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = at[kk * ACfg::LDT + i * 32];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = bt[kk * BCfg::LDT + j * 32];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      if (p.rowsum && !A_KMAJOR) {
#pragma unroll
        for (int v = 0; v < ACfg::PER_THREAD; ++v) rs[v] += (ra[v].x + ra[v].y) + (ra[v].z + ra[v].w);
      }
      tile_store<BM, A_KMAJOR>(ra, lds + (stage ^ 1) * STAGE_FLOATS, tid);
      tile_store<BN, B_KMAJOR>(rb, lds + (stage ^ 1) * STAGE_FLOATS + ACfg::LDS_FLOATS, tid);
    }
    __syncthreads();
    stage ^= 1;
    have = more;
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* C = p.C + (size_t)z * p.c_bs;
  const float scale = p.scale ? p.scale[g / p.scale_div] : 1.f;
  const float* mul = p.mul ? p.mul + (size_t)(g / p.mul_div) * p.mul_bs : nullptr;
  const float* dact = p.dact ? p.dact + (size_t)g * p.dact_bs : nullptr;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (!wave_ok) continue;
      const int n = n0 + wn * TN + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < p.M && n < p.N) {
          float v = acc[i][j][r] * scale;
          if (p.bias) v += p.bias[m];
          if (p.act == 1) v = v > 0.f ? v : kSlope * v;
          if (mul) v *= mul[(size_t)m * p.mul_ld + n];
          const size_t off = (size_t)m * p.ldc + n;
          if (dact) v *= dact[off] > 0.f ? 1.f : kSlope;
          if (p.accumulate) v += C[off];
          C[off] = v;
        } else if (m < p.M && n < nstore) {
          C[(size_t)m * p.ldc + n] = 0.f;
        }
      }
    }

  if (p.rowsum && !A_KMAJOR && blockIdx.x == 0) {
    // rows of the A slab are split over 4 consecutive lanes (BK/4 float4 per row)
#pragma unroll
    for (int v = 0; v < ACfg::PER_THREAD; ++v) {
      float t = rs[v];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      const int idx = tid + v * 256;
      const int m = m0 + idx / (BK / 4);
      if ((idx & 3) == 0 && (ACfg::NVEC % 256 == 0 || idx < ACfg::NVEC) && m < p.M) p.rowsum[(size_t)z * p.M + m] = t;
    }
  }
}

template <int BM, int BN, int WM, int WN>
inline int launch_gemm_cfg(const GemmArgs& a, int batch_z, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  dim3 grid(cdiv(a.n_store > a.N ? a.n_store : a.N, BN), cdiv(a.M, BM), batch_z), block(256);
  const int tag = a.prof_class;
#define DPPM_GEMM_LAUNCH(AK, BKM, TAG) \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_f32_kernel<BM, BN, WM, WN, AK, BKM, TAG>), grid, block, 0, st, a)
  if (tag == PROF_LINEAR) {
    if (!a_kmajor && b_kmajor) DPPM_GEMM_LAUNCH(false, true, 0);
    else if (a_kmajor && b_kmajor) DPPM_GEMM_LAUNCH(true, true, 0);
    else return fail(DPPM_E_UNSUPPORTED, "gemm(linear): operand layout not instantiated");
  } else if (tag == PROF_PLANE_MM) {
    if (!a_kmajor && b_kmajor) DPPM_GEMM_LAUNCH(false, true, 1);
    else if (a_kmajor && b_kmajor) DPPM_GEMM_LAUNCH(true, true, 1);
    else if (!a_kmajor && !b_kmajor) DPPM_GEMM_LAUNCH(false, false, 1);
    else return fail(DPPM_E_UNSUPPORTED, "gemm(plane): A^T * B^T is not instantiated");
  } else if (tag == PROF_WGRAD) {
    if (!a_kmajor && !b_kmajor) DPPM_GEMM_LAUNCH(false, false, 2);
    else return fail(DPPM_E_UNSUPPORTED, "gemm(wgrad): operand layout not instantiated");
  } else {
    return fail(DPPM_E_INVALID, "gemm: unknown class %d", tag);
  }
#undef DPPM_GEMM_LAUNCH
  return check_launch("gemm_f32");
}

// batch_z = number of z slices (batch * nsplit)
inline int launch_gemm(const GemmArgs& a, int batch_z, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || batch_z <= 0) return fail(DPPM_E_INVALID, "gemm: empty problem M=%d N=%d z=%d", a.M, a.N, batch_z);
  if (batch_z > 65535) return fail(DPPM_E_INVALID, "gemm: batch %d exceeds grid.z limit", batch_z);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (a.M > 64) rc = launch_gemm_cfg<128, 128, 2, 2>(a, batch_z, a_kmajor, b_kmajor, st);
  else if (a.M > 32) rc = launch_gemm_cfg<64, 128, 2, 2>(a, batch_z, a_kmajor, b_kmajor, st);
  else rc = launch_gemm_cfg<32, 128, 1, 4>(a, batch_z, a_kmajor, b_kmajor, st);
  if (prof) {
    double k = 0;
    for (int i = 0; i < a.nseg; ++i) k += a.seg[i].K;
    const double batch = a.nsplit > 1 ? (double)batch_z / a.nsplit : (double)batch_z;
    prof_after(st, a.prof_class, AR_F32, 2.0 * a.M * a.N * k * batch);
  }
  return rc;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace dppm
