// Per-channel N x N contraction C[g] = A[g] * B[g] * scale (model/ppgn.py:62-64, forward form) for 129 <= N <= 1024 on
// the bf16 matrix pipe with split operands (see dppm_wsplit.h: x = hi + lo, product = hi*hi + hi*lo + lo*hi, float32
// accumulation; NS = 1 keeps the hi pieces only).  Operands and result stay float32 in HBM: the kernel is HBM-bound
// (3 planes of N*ld floats per product against 3 * 2 N^3 / 16 f32-MFMA-equivalents of matrix time).
//
// One 16-wave workgroup per plane (N <= 256; larger planes are cut into 256 x 256 output blocks, one workgroup each, K = N),
// waves in the 4 x 4 Latin-square cut of the 16 x 16 output tiles of dppm_planemm.h.
// K advances 32 at a time.  Per panel two bf16 images (hi, lo) of each operand live in LDS, both as [row][32 k] with k
// contiguous (64-byte rows), so every MFMA fragment (8 consecutive k of one row) is one ds_read_b128:
//   A image: row = m, straight from A[m][k] (waves 8-15 stage it: float4 along k -> 8 bytes);
//   B image: row = n, i.e. B TRANSPOSED while staging (waves 0-7: four float4 along n from rows k .. k+3 -> for each n
//            4 consecutive k = 8 bytes).
// 16-byte chunk c (0..3) of row r sits at chunk c ^ ((-(r >> 2)) & 3): conflict-free for the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, ...) and for the A-side stores; the transposing B-side stores are 4-way conflicted (the kernel has
// a 3x margin under its HBM time).  The MFMA is issued with the operands swapped (D = B^T-frag x A-frag), so a lane ends
// up with 4 consecutive columns of one row of C: 16-byte stores (plane_store_tile).
// The next panel's global loads are issued before the MFMAs of the current one and landed after them in the other LDS
// buffer: one barrier per panel.
#pragma once
#include "dppm_planemm.h"
#include "dppm_wsplit.h"

namespace dppm {

constexpr int P3_BK = 32;

template <int NT>
struct P3Cfg {
  static constexpr int ROWS = NT * 16;
  static constexpr int IMG = ROWS * 64;          // bytes of one bf16 image of one operand panel
};

__device__ __forceinline__ unsigned p3_off(unsigned row, unsigned c) { return row * 64 + ((c ^ ((0u - (row >> 2)) & 3u)) << 4); }

// prefetch set of one staging thread: 4 float4 (named members: stays in registers)
struct P3Pre { float4 a, b, c, d; };

template <int NS>
__device__ __forceinline__ void p3_put(char* __restrict__ img, unsigned off, int img_bytes, float x0, float x1, float x2, float x3) {
  unsigned h01, l01 = 0, h23, l23 = 0;
  w3_split2<NS>(x0, x1, h01, l01);
  w3_split2<NS>(x2, x3, h23, l23);
  *reinterpret_cast<uint2*>(img + off) = make_uint2(h01, h23);
  if (NS > 1) *reinterpret_cast<uint2*>(img + img_bytes + off) = make_uint2(l01, l23);
}

// the whole product for a wave that owns NR row tiles (from r0) x NC column tiles (from c0)
// One finished 16 x 16 tile of an output block (held transposed, see plane_store_tile): rows < nrows, columns < ncols are
// results, columns in [ncols, wcols) are written as zeros (wcols = readable / writable width of the block's rows).
__device__ __forceinline__ void p3_store_tile(const floatx4& acc, float* __restrict__ C, const float* __restrict__ mask, float scale,
                                              int row0, int col0, int nrows, int ncols, int wcols, int ld, int lane) {
  const int row = row0 + (lane & 15), col = col0 + 4 * (lane >> 4);
  if (row < nrows && col < wcols) {
    float4 v = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
    if (mask) {
      const float4 m = *reinterpret_cast<const float4*>(mask + (size_t)row * ld + col);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
    const int lim = ncols - col;
    if (lim < 1) v.x = 0.f;
    if (lim < 2) v.y = 0.f;
    if (lim < 3) v.z = 0.f;
    if (lim < 4) v.w = 0.f;
    *reinterpret_cast<float4*>(C + (size_t)row * ld + col) = v;
  }
}

// (mlo, nlo): first row / column of this workgroup's output block; nrows / ncols: its valid extent; K = p.n
template <int NT, int NS, int NR, int NC>
__device__ __forceinline__ void p3_body(const PlaneArgs& p, char* lds, int g, int mlo, int nlo, int r0, int c0, int tid, int lane) {
  using Cfg = P3Cfg<NT>;
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int OPB = NIMG * Cfg::IMG;           // bytes of one operand (its images) in one stage
  constexpr int STAGE = 2 * OPB;                 // A images then B images
  const size_t plane = (size_t)p.n * p.ld;
  const int n = p.n, ld = p.ld;
  const int nrows = n - mlo < Cfg::ROWS ? n - mlo : Cfg::ROWS, ncols = n - nlo < Cfg::ROWS ? n - nlo : Cfg::ROWS;
  const int wcols = ld - nlo < Cfg::ROWS ? ld - nlo : Cfg::ROWS;   // ld % 4 == 0 and nlo % 256 == 0: whole float4s
  const float* A = p.A + g * plane + (size_t)mlo * ld;
  const float* B = p.B + g * plane + nlo;
  const bool stage_b = tid < 512;
  const int u = tid & 511;
  // B side: kq = u >> 6 (k = 4 kq ..), columns 4 nq .. 4 nq + 3, nq = u & 63.   A side: kq = u & 7, rows (u >> 3) + 64 v.
  const int kq = stage_b ? (u >> 6) : (u & 7);
  const int nq = u & 63, mrow = u >> 3;

  floatx4 acc[NR > 0 ? NR : 1][NC > 0 ? NC : 1];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // unconditional, clamped loads (see dppm_wsplit.h): out-of-range threads re-read element 0 and store zeros
  auto issue = [&](int k0, P3Pre& q) {
    if (stage_b) {
      const int k = k0 + 4 * kq;
      const bool okc = 4 * nq < wcols;
      const float* b0 = B + (okc ? 4 * nq : 0);
      q.a = *reinterpret_cast<const float4*>(b0 + (size_t)(k + 0 < n ? k + 0 : 0) * ld);
      q.b = *reinterpret_cast<const float4*>(b0 + (size_t)(k + 1 < n ? k + 1 : 0) * ld);
      q.c = *reinterpret_cast<const float4*>(b0 + (size_t)(k + 2 < n ? k + 2 : 0) * ld);
      q.d = *reinterpret_cast<const float4*>(b0 + (size_t)(k + 3 < n ? k + 3 : 0) * ld);
    } else {
      const int k = k0 + 4 * kq;
      const float* a0 = A + (k < ld ? k : 0);
      q.a = *reinterpret_cast<const float4*>(a0 + (size_t)(mrow < nrows ? mrow : 0) * ld);
      q.b = *reinterpret_cast<const float4*>(a0 + (size_t)(mrow + 64 < nrows ? mrow + 64 : 0) * ld);
      q.c = *reinterpret_cast<const float4*>(a0 + (size_t)(mrow + 128 < nrows ? mrow + 128 : 0) * ld);
      q.d = *reinterpret_cast<const float4*>(a0 + (size_t)(mrow + 192 < nrows ? mrow + 192 : 0) * ld);
    }
  };
  auto land = [&](int k0, const P3Pre& q, char* stage) {
    const int k = k0 + 4 * kq;
    const bool k0ok = k + 0 < n, k1ok = k + 1 < n, k2ok = k + 2 < n, k3ok = k + 3 < n;
    const unsigned cb = (kq >> 1), hb = (kq & 1) * 8;
    if (stage_b) {   // element (k + v, nlo + 4 nq + e) of B -> image row 4 nq + e; zero when k + v >= n or the column is outside
      char* img = stage + OPB;
      auto col_put = [&](int row, float x0, float x1, float x2, float x3) {
        if (row < Cfg::ROWS) {
          const bool r = row < ncols;
          p3_put<NS>(img, p3_off(row, cb) + hb, Cfg::IMG, r && k0ok ? x0 : 0.f, r && k1ok ? x1 : 0.f, r && k2ok ? x2 : 0.f, r && k3ok ? x3 : 0.f);
        }
      };
      col_put(4 * nq + 0, q.a.x, q.b.x, q.c.x, q.d.x);
      col_put(4 * nq + 1, q.a.y, q.b.y, q.c.y, q.d.y);
      col_put(4 * nq + 2, q.a.z, q.b.z, q.c.z, q.d.z);
      col_put(4 * nq + 3, q.a.w, q.b.w, q.c.w, q.d.w);
    } else {         // element (row, k + v) of A -> image row `row`
      char* img = stage;
      auto row_put = [&](int row, const float4& v) {
        if (row < Cfg::ROWS) {
          const bool r = row < nrows;
          p3_put<NS>(img, p3_off(row, cb) + hb, Cfg::IMG, r && k0ok ? v.x : 0.f, r && k1ok ? v.y : 0.f, r && k2ok ? v.z : 0.f, r && k3ok ? v.w : 0.f);
        }
      };
      row_put(mrow, q.a); row_put(mrow + 64, q.b); row_put(mrow + 128, q.c); row_put(mrow + 192, q.d);
    }
  };

  P3Pre pre;
  issue(0, pre);
  land(0, pre, lds);
  __syncthreads();
  const int ci = lane & 15, gq = lane >> 4;
  int stage = 0;
  auto mma = [&](int st) {
    if (NR > 0 && NC > 0) {
      const char* ai = lds + st * STAGE;
      const char* bi = ai + OPB;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const unsigned ao = p3_off((r0 + i) * 16 + ci, gq);
        const uintx4 ah = *reinterpret_cast<const uintx4*>(ai + ao);
        uintx4 al = ah;
        if (NS > 1) al = *reinterpret_cast<const uintx4*>(ai + Cfg::IMG + ao);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const unsigned bo = p3_off((c0 + j) * 16 + ci, gq);
          const uintx4 bh = *reinterpret_cast<const uintx4*>(bi + bo);
          if (NS > 1) {
            const uintx4 bl = *reinterpret_cast<const uintx4*>(bi + Cfg::IMG + bo);
            acc[i][j] = w3_mfma(bl, ah, acc[i][j]);
            acc[i][j] = w3_mfma(bh, al, acc[i][j]);
          }
          acc[i][j] = w3_mfma(bh, ah, acc[i][j]);
        }
      }
    }
  };
  int k0 = 0;
  for (; k0 + P3_BK < n; k0 += P3_BK) {   // every panel but the last: request the next one, multiply, land it
    issue(k0 + P3_BK, pre);
    mma(stage);
    land(k0 + P3_BK, pre, lds + (stage ^ 1) * STAGE);
    __syncthreads();
    stage ^= 1;
  }
  mma(stage);                             // last panel: nothing left to request (no wasted re-read)
  const float scale = p.scale ? p.scale[g / p.div] : 1.f;
  const size_t boff = (size_t)mlo * ld + nlo;
  const float* mask = p.mask ? p.mask + (size_t)(g / p.div) * plane + boff : nullptr;
  float* C = p.C + g * plane + boff;
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) p3_store_tile(acc[i][j], C, mask, scale, (r0 + i) * 16, (c0 + j) * 16, nrows, ncols, wcols, ld, lane);
}

template <int NT, int NS>
__global__ __launch_bounds__(1024) void plane3_kernel(const PlaneArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  constexpr int BASE = NT / 4, REM = NT % 4;       // group q has BASE + (q < REM) members and starts at q*BASE + min(q, REM)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = wave >> 2, b = wave & 3;
  const int cg = (a + b) & 3;
  const int r0 = a * BASE + (a < REM ? a : REM), c0 = cg * BASE + (cg < REM ? cg : REM);
  const bool rbig = a < REM, cbig = cg < REM;
  const int g = blockIdx.x;
  const int mlo = (int)blockIdx.y * P3Cfg<NT>::ROWS, nlo = (int)blockIdx.z * P3Cfg<NT>::ROWS;   // (0, 0) when the plane is one block
  if (REM == 0) p3_body<NT, NS, BASE, BASE>(p, lds3, g, mlo, nlo, r0, c0, tid, lane);
  else if (rbig && cbig) p3_body<NT, NS, BASE + 1, BASE + 1>(p, lds3, g, mlo, nlo, r0, c0, tid, lane);
  else if (rbig) p3_body<NT, NS, BASE + 1, BASE>(p, lds3, g, mlo, nlo, r0, c0, tid, lane);
  else if (cbig) p3_body<NT, NS, BASE, BASE + 1>(p, lds3, g, mlo, nlo, r0, c0, tid, lane);
  else p3_body<NT, NS, BASE, BASE>(p, lds3, g, mlo, nlo, r0, c0, tid, lane);
}

template <int NT, int NS>
inline int plane3_launch_nt(const PlaneArgs& a, int batch, int nblk, hipStream_t st) {
  const size_t lds = (size_t)2 * 2 * (NS > 1 ? 2 : 1) * P3Cfg<NT>::IMG;   // stages x operands x images
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(&plane3_kernel<NT, NS>), (size_t)(lds), "plane3"));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(plane3_kernel<NT, NS>), dim3(batch, nblk, nblk), dim3(1024), lds, st, a);
  return check_launch("plane3");
}

inline bool plane3_supported(int n) { return n > 128 && n <= 1024; }

// forward form only: C = A * B (A, B row-major planes [n][ld])
template <int NS>
inline int plane3_launch(const PlaneArgs& a, int batch, hipStream_t st) {
  const int nt = cdiv(a.n, 16);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (cdiv(a.n, 256) > 4) rc = fail(DPPM_E_UNSUPPORTED, "plane3: n=%d outside 129..1024", a.n);
  else if (nt > 16) rc = plane3_launch_nt<16, NS>(a, batch, cdiv(a.n, 256), st);   // 256 x 256 output blocks
  else switch (nt) {
    case 9: rc = plane3_launch_nt<9, NS>(a, batch, 1, st); break;
    case 10: rc = plane3_launch_nt<10, NS>(a, batch, 1, st); break;
    case 11: rc = plane3_launch_nt<11, NS>(a, batch, 1, st); break;
    case 12: rc = plane3_launch_nt<12, NS>(a, batch, 1, st); break;
    case 13: rc = plane3_launch_nt<13, NS>(a, batch, 1, st); break;
    case 14: rc = plane3_launch_nt<14, NS>(a, batch, 1, st); break;
    case 15: rc = plane3_launch_nt<15, NS>(a, batch, 1, st); break;
    case 16: rc = plane3_launch_nt<16, NS>(a, batch, 1, st); break;
    default: rc = fail(DPPM_E_UNSUPPORTED, "plane3: n=%d outside 129..1024", a.n);
  }
  if (prof) prof_after(st, PROF_PLANE_MM, prof_arith_of_pieces(NS), 2.0 * a.n * a.n * (double)a.n * batch, 3.0 * sizeof(float) * a.n * (double)a.ld * batch);
  return rc;
}

}  // namespace dppm
