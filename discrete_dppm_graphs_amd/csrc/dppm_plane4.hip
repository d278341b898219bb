// Per-channel N x N contraction C[g] = A[g] * B[g] * scale (model/ppgn.py:62-64), split-bf16 arithmetic, 129 <= N <= 512:
// the role-split, persistent form of dppm_planesplit.h (same arithmetic in the same order: same results bit for bit).
//
// plane3_kernel lets all 16 waves stage a K = 32 panel AND multiply the previous one: one panel (53 KB) in flight per CU, and
// only between its request and its landing -- at ~2 us of HBM latency under load that duty cycle, not the matrix pipe (19 %
// busy, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES) or the LDS (17 %), set its 3.6 TB/s.  Here one 12-wave workgroup per CU walks its
// planes as ONE stream of panels:
//   waves 0-3  (stagers)     keep two panels in flight in registers beside the one they are landing (split into bf16 hi / lo
//                            images, B transposed on the way); they never wait for a multiplier except at the panel barrier
//                            and run ahead across plane boundaries;
//   waves 4-11 (multipliers) own the 16 x 16 output tiles in a 4 (row groups) x 2 (column halves) cut -- up to 4 x 7 tiles in
//                            112 accumulator registers at 3 waves per SIMD (168 registers) -- read fragments from LDS, and at a
//                            plane's last panel scale and store their tiles with 16-byte stores while the stagers go on.
// One barrier per panel; LDS: two panel stages (106 KB at N = 200).
//
// NS = 6 (training, "float32" precision): three bf16 pieces per operand (x = hi + mid + lo, 24 mantissa bits) and the six products
// down to 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; smallest first, float32 accumulation) -- the split that
// dppm_wgrad.h uses for the weight gradients -- at 6/16 of the matrix time of v_mfma_f32_16x16x4_f32, which the exact-f32
// kernel of dppm_planemm.h spends 48 % of its peak on.  The backward forms come with it: dA = dC B^T (TB) and dB = A^T dC (TA)
// only change which operand is copied straight (k along its rows) and which is transposed while staging.  Three images per
// operand fill the LDS at N <= 208 (2 x 2 x 3 x 13 KB = 156 KB); larger planes keep the exact-f32 kernel.
#include "dppm_plane4.h"
#include "dppm_planesplit.h"

namespace dppm {

constexpr int P4_THREADS = 768;
constexpr int P4_STAGERS = 256;
constexpr int P4_CUS = 256;                                   // MI355X: 8 XCDs x 32 CUs

template <int NT>
struct P4Cfg {
  static constexpr int ROWS = NT * 16;
  static constexpr int IMG = ROWS * 64;                       // bytes of one bf16 image of one operand panel
  static constexpr int NQ = ROWS / 4;                         // float4 columns of a B panel row
  static constexpr int VA = (ROWS * 8 + P4_STAGERS - 1) / P4_STAGERS;     // float4 per staging thread, A side: (row, 4 k)
  static constexpr int VB = (8 * NQ + P4_STAGERS - 1) / P4_STAGERS;       // quads per staging thread, B side: (4 k rows, 4 columns)
};

// The workgroup's share: planes group + i * ngroups, output block (mlo, nlo) of each.  NBLK = 2 (planes cut into 2 x 2 blocks of
// 16 NT rows, K = n): the four workgroups blockIdx.x = x + 8 (4 s + member), member = 0..3, sit on one XCD (workgroups go to XCDs
// round-robin) and walk the same planes in step, so each half of A and of B is fetched from HBM once and found in that XCD's L2
// by its second reader.
struct P4Work { int group, ngroups, mlo, nlo; };
template <int NBLK, int ROWS>
__device__ __forceinline__ P4Work p4_work() {
  P4Work w;
  if (NBLK == 1) { w.group = blockIdx.x; w.ngroups = gridDim.x; w.mlo = 0; w.nlo = 0; }
  else {
    const int x = blockIdx.x, member = (x >> 3) & 3;
    w.group = (x & 7) + 8 * (x >> 5); w.ngroups = gridDim.x >> 2;
    w.mlo = (member >> 1) * ROWS; w.nlo = (member & 1) * ROWS;
  }
  return w;
}

// Image layout [row][32 k] of dppm_planesplit.h (64-byte rows, 16-byte chunk c of row r at chunk c ^ ((-(r >> 2)) & 3)) with one
// change: rows r and r ^ 1 trade places where bit 4 of r is set.  A transposing staging thread writes 8 bytes (4 consecutive k) of
// rows 4 nq + c; with the thread's k quad parity in lane bit 0 and nq in the bits above, the 16 lanes of a ds_write_b64 group
// differ in the 8-byte half (k quad parity), the chunk (nq & 3) and the 64-byte half of the 128-byte bank span (row bit 0 ^ nq
// bit 2): 16 distinct slots, conflict-free (it was 4-way: 31 % of plane3's LDS cycles were conflict cycles).  Fragment reads see
// rows ci ^ (tile & 1) of tile `tile`: the same address set per ds_read_b128 lane group, still conflict-free.
__device__ __forceinline__ unsigned p4_off(unsigned row, unsigned c) { return ((row ^ ((row >> 4) & 1u)) * 64) + ((c ^ ((0u - (row >> 2)) & 3u)) << 4); }

template <int NX, int NY>
struct P4Pre { float4 x[NX]; float4 y[NY]; };   // one panel in flight per staging thread: A-side and B-side pieces

template <int NS> struct P4Img { static constexpr int N = NS == 1 ? 1 : NS == 3 ? 2 : 3; };

// four consecutive k of one image row -> 8 bytes in each piece image
template <int NS>
__device__ __forceinline__ void p4_put(char* __restrict__ img, unsigned off, int img_bytes, float x0, float x1, float x2, float x3) {
  if (NS == 6) {
    const unsigned h01 = w3_pack(x0, x1), h23 = w3_pack(x2, x3);
    const float r0 = x0 - __uint_as_float(h01 << 16), r1 = x1 - __uint_as_float(h01 & 0xffff0000u);
    const float r2 = x2 - __uint_as_float(h23 << 16), r3 = x3 - __uint_as_float(h23 & 0xffff0000u);
    const unsigned m01 = w3_pack(r0, r1), m23 = w3_pack(r2, r3);
    const unsigned l01 = w3_pack(r0 - __uint_as_float(m01 << 16), r1 - __uint_as_float(m01 & 0xffff0000u));
    const unsigned l23 = w3_pack(r2 - __uint_as_float(m23 << 16), r3 - __uint_as_float(m23 & 0xffff0000u));
    *reinterpret_cast<uint2*>(img + off) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(img + img_bytes + off) = make_uint2(m01, m23);
    *reinterpret_cast<uint2*>(img + 2 * img_bytes + off) = make_uint2(l01, l23);
  } else {
    p3_put<NS>(img, off, img_bytes, x0, x1, x2, x3);
  }
}

template <int NT, int NS, int NR, int NC>
__device__ __forceinline__ void p4_multiply(const PlaneArgs& p, const char* lds, int r0, int c0, int lane, int nloc, int np, int dbg, const P4Work& wk) {
  using Cfg = P4Cfg<NT>;
  constexpr int NIMG = P4Img<NS>::N;
  constexpr int OPB = NIMG * Cfg::IMG, STAGE = 2 * OPB;
  const int n = p.n, ld = p.ld;
  const size_t plane = (size_t)n * ld;
  const int nrows = n - wk.mlo < Cfg::ROWS ? n - wk.mlo : Cfg::ROWS, ncols = n - wk.nlo < Cfg::ROWS ? n - wk.nlo : Cfg::ROWS;
  const int wcols = ld - wk.nlo < Cfg::ROWS ? ld - wk.nlo : Cfg::ROWS;   // ld % 4 == 0, nlo % 16 == 0: whole float4s
  const size_t boff = (size_t)wk.mlo * ld + wk.nlo;
  const int ci = lane & 15, gq = lane >> 4;
  // p4_off(16 t + ci, gq) = 1024 t + p4_off(ci ^ (t & 1), gq): two lane offsets (even / odd tiles), tile offsets as immediates
  const unsigned lo_e = p4_off(ci, gq), lo_o = p4_off(ci ^ 1, gq);
  const unsigned abase0 = ((r0 & 1) ? lo_o : lo_e) + r0 * 1024, abase1 = ((r0 & 1) ? lo_e : lo_o) + r0 * 1024;   // tiles r0 + even / odd
  const unsigned bbase0 = OPB + ((c0 & 1) ? lo_o : lo_e) + c0 * 1024, bbase1 = OPB + ((c0 & 1) ? lo_e : lo_o) + c0 * 1024;
  floatx4 acc[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  // A finished plane leaves in one burst after its last panel (measured: spreading it over two intervals -- half the row tiles
  // now, half before the next plane's first MFMAs -- was 5 % slower; streaming "nt" stores 40 % slower: the L2 merges the 64-byte
  // row pieces of neighbouring tiles; a three-stage LDS ring with per-slot landed / consumed counters instead of the panel
  // barrier, which lets the stagers keep landing and requesting during the burst, was bit-identical and 8 % slower: loads
  // queue behind the burst in the CU's memory pipeline whoever issues them -- a CU moves ~19 GB/s in either direction here).
  auto store_plane = [&](int i) {
    const long g = (long)wk.group + (long)i * wk.ngroups;
    const float scale = p.scale ? p.scale[g / p.div] : 1.f;
    const float* mask = p.mask ? p.mask + (size_t)(g / p.div) * plane + boff : nullptr;
    float* C = p.C + g * plane + boff;
    if (NS == 6 && mask) {   // backward forms: the mask pieces of two tiles are requested before the first of them is used
      constexpr int MG = 2;
#pragma unroll
      for (int ii = 0; ii < NR; ++ii) {
        const int row = (r0 + ii) * 16 + (lane & 15);
#pragma unroll
        for (int j0 = 0; j0 < NC; j0 += MG) {
          float4 mk[MG];
#pragma unroll
          for (int jj = 0; jj < MG; ++jj) {
            const int col = (c0 + j0 + jj) * 16 + 4 * (lane >> 4);
            if (j0 + jj < NC) mk[jj] = *reinterpret_cast<const float4*>(mask + (size_t)(row < nrows ? row : 0) * ld + (col < wcols ? col : 0));
          }
#pragma unroll
          for (int jj = 0; jj < MG; ++jj) {
            if (j0 + jj < NC) {
              const int j = j0 + jj;
              const int col = (c0 + j) * 16 + 4 * (lane >> 4);
              if (row < nrows && col < wcols) {
                float4 v = make_float4(acc[ii][j][0] * scale * mk[jj].x, acc[ii][j][1] * scale * mk[jj].y, acc[ii][j][2] * scale * mk[jj].z, acc[ii][j][3] * scale * mk[jj].w);
                const int lim = ncols - col;
                if (lim < 1) v.x = 0.f;
                if (lim < 2) v.y = 0.f;
                if (lim < 3) v.z = 0.f;
                if (lim < 4) v.w = 0.f;
                *reinterpret_cast<float4*>(C + (size_t)row * ld + col) = v;
              }
              acc[ii][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int ii = 0; ii < NR; ++ii)
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        p3_store_tile(acc[ii][j], C, mask, scale, (r0 + ii) * 16, (c0 + j) * 16, nrows, ncols, wcols, ld, lane);
        acc[ii][j] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
  };
  __syncthreads();                                            // panel 0 is in stage 0
  int q = 0;
  for (int i = 0; i < nloc; ++i) {
    for (int kp = 0; kp < np; ++kp, ++q) {
      const char* st = lds + (q & 1) * STAGE;
      if (!(dbg & 1)) {
        // fragments one (row tile, column tile) step ahead of the MFMAs that use them; the order is pinned (sched_group_barrier) so
        // that the scheduler does not hoist a panel's worth of reads over the accumulators' registers.  (In the ISA every group of
        // reads still sits right in front of the MFMAs that consume it.  Two B fragment sets used in turn with sched_barrier(0)
        // between the steps do put the reads a full step ahead -- and change nothing: N = 200 0.479 vs 0.477 ms, N = 512 8.21 vs
        // 8.23 ms, f32x6 0.643 vs 0.640 ms.  The multipliers are not what this kernel waits for.)
        constexpr bool AHEAD = NR * NC <= 28 && NS != 6;      // 32 tiles, three pieces: no registers left for a second fragment set
        if (NS == 6) {
#pragma unroll
        for (int ii = 0; ii < NR; ++ii) {
          const uintx4 ah = *reinterpret_cast<const uintx4*>(st + ((ii & 1) ? abase1 : abase0) + ii * 1024);
          const uintx4 am = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((ii & 1) ? abase1 : abase0) + ii * 1024);
          const uintx4 al = *reinterpret_cast<const uintx4*>(st + 2 * Cfg::IMG + ((ii & 1) ? abase1 : abase0) + ii * 1024);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const uintx4 bh = *reinterpret_cast<const uintx4*>(st + ((j & 1) ? bbase1 : bbase0) + j * 1024);
            const uintx4 bm = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((j & 1) ? bbase1 : bbase0) + j * 1024);
            const uintx4 bl = *reinterpret_cast<const uintx4*>(st + 2 * Cfg::IMG + ((j & 1) ? bbase1 : bbase0) + j * 1024);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            acc[ii][j] = w3_mfma(bm, am, acc[ii][j]);             // smallest terms first
            acc[ii][j] = w3_mfma(bh, al, acc[ii][j]);
            acc[ii][j] = w3_mfma(bl, ah, acc[ii][j]);
            acc[ii][j] = w3_mfma(bh, am, acc[ii][j]);
            acc[ii][j] = w3_mfma(bm, ah, acc[ii][j]);
            acc[ii][j] = w3_mfma(bh, ah, acc[ii][j]);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          }
        }
        } else if (AHEAD) {
        uintx4 ah = *reinterpret_cast<const uintx4*>(st + abase0), al = ah;
        if (NS > 1) al = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + abase0);
        uintx4 bh = *reinterpret_cast<const uintx4*>(st + bbase0), bl = bh;
        if (NS > 1) bl = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + bbase0);
#pragma unroll
        for (int t = 0; t < NR * NC; ++t) {
          const int ii = t / NC, j = t % NC;
          const uintx4 cah = ah, cal = al, cbh = bh, cbl = bl;
          if (t + 1 < NR * NC) {
            const int in = (t + 1) / NC, jn = (t + 1) % NC;
            bh = *reinterpret_cast<const uintx4*>(st + ((jn & 1) ? bbase1 : bbase0) + jn * 1024);
            if (NS > 1) bl = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((jn & 1) ? bbase1 : bbase0) + jn * 1024);
            if (jn == 0) {
              ah = *reinterpret_cast<const uintx4*>(st + ((in & 1) ? abase1 : abase0) + in * 1024);
              if (NS > 1) al = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((in & 1) ? abase1 : abase0) + in * 1024);
            }
            if (jn == 0) __builtin_amdgcn_sched_group_barrier(0x100, NS > 1 ? 4 : 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, NS > 1 ? 2 : 1, 0);
          }
          if (NS > 1) {
            acc[ii][j] = w3_mfma(cbl, cah, acc[ii][j]);
            acc[ii][j] = w3_mfma(cbh, cal, acc[ii][j]);
          }
          acc[ii][j] = w3_mfma(cbh, cah, acc[ii][j]);
          __builtin_amdgcn_sched_group_barrier(0x008, NS > 1 ? 3 : 1, 0);
        }
        } else {
#pragma unroll
        for (int ii = 0; ii < NR; ++ii) {
          const uintx4 ah = *reinterpret_cast<const uintx4*>(st + ((ii & 1) ? abase1 : abase0) + ii * 1024);
          uintx4 al = ah;
          if (NS > 1) al = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((ii & 1) ? abase1 : abase0) + ii * 1024);
          __builtin_amdgcn_sched_group_barrier(0x100, NS > 1 ? 2 : 1, 0);
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const uintx4 bh = *reinterpret_cast<const uintx4*>(st + ((j & 1) ? bbase1 : bbase0) + j * 1024);
            uintx4 bl = bh;
            if (NS > 1) bl = *reinterpret_cast<const uintx4*>(st + Cfg::IMG + ((j & 1) ? bbase1 : bbase0) + j * 1024);
            __builtin_amdgcn_sched_group_barrier(0x100, NS > 1 ? 2 : 1, 0);
            if (NS > 1) {
              acc[ii][j] = w3_mfma(bl, ah, acc[ii][j]);
              acc[ii][j] = w3_mfma(bh, al, acc[ii][j]);
            }
            acc[ii][j] = w3_mfma(bh, ah, acc[ii][j]);
            __builtin_amdgcn_sched_group_barrier(0x008, NS > 1 ? 3 : 1, 0);
          }
        }
        }
      }
      if (kp == np - 1 && !(dbg & 8)) store_plane(i);
      __syncthreads();
    }
  }
}

template <int NT, int NS, int NBLK, bool TA, bool TB>
__global__ __launch_bounds__(P4_THREADS) void plane4_kernel(const PlaneArgs p, int batch, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char lds4[];
  using Cfg = P4Cfg<NT>;
  constexpr int NIMG = P4Img<NS>::N;
  constexpr int OPB = NIMG * Cfg::IMG, STAGE = 2 * OPB;
  constexpr int VA = Cfg::VA, VB = Cfg::VB, NQ = Cfg::NQ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = p.n, ld = p.ld;
  const int np = (n + P3_BK - 1) / P3_BK;
  const P4Work wk = p4_work<NBLK, Cfg::ROWS>();
  if (wk.group >= batch || wk.mlo >= n || wk.nlo >= n) return;   // whole workgroup, before any barrier
  const int nloc = (batch - wk.group + wk.ngroups - 1) / wk.ngroups;   // planes group + i * ngroups
  const int nq = nloc * np;                                   // panels of this workgroup, one stream

  if (wave >= P4_STAGERS / 64) {
    // ---------------------------------------------------------------------------------------------- multipliers
    constexpr int BASE = NT / 4, REM = NT % 4;                // row group a: BASE + (a < REM) tiles
    constexpr int CH0 = (NT + 1) / 2, CH1 = NT / 2;           // column halves
    const int m = wave - P4_STAGERS / 64, a = m >> 1, b = m & 1;
    const int r0 = a * BASE + (a < REM ? a : REM), c0 = b ? CH0 : 0;
    const bool rbig = a < REM;
    if (CH0 == CH1) {
      if (REM != 0 && rbig) p4_multiply<NT, NS, BASE + 1, CH0>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
      else p4_multiply<NT, NS, BASE, CH0>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
    } else {
      if (REM != 0 && rbig && b == 0) p4_multiply<NT, NS, BASE + 1, CH0>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
      else if (REM != 0 && rbig) p4_multiply<NT, NS, BASE + 1, CH1>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
      else if (b == 0) p4_multiply<NT, NS, BASE, CH0>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
      else p4_multiply<NT, NS, BASE, CH1>(p, lds4, r0, c0, lane, nloc, np, dbg, wk);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- stagers
  // An operand is staged either STRAIGHT (image row r = its matrix row, k along that row: float4 along k -> 8 bytes) or
  // TRANSPOSING (image row = its matrix column, k along the matrix rows: four float4 from rows k .. k+3 -> for each of 4
  // columns 4 consecutive k = 8 bytes).  A: straight unless TA (C = A^T B); B: transposing unless TB (C = A B^T).
  const size_t plane = (size_t)n * ld;
  const int nrows = n - wk.mlo < Cfg::ROWS ? n - wk.mlo : Cfg::ROWS, ncols = n - wk.nlo < Cfg::ROWS ? n - wk.nlo : Cfg::ROWS;
  const int wrows = ld - wk.mlo < Cfg::ROWS ? ld - wk.mlo : Cfg::ROWS, wcols = ld - wk.nlo < Cfg::ROWS ? ld - wk.nlo : Cfg::ROWS;
  const int skq = tid & 7, srow = tid >> 3;                   // straight: k = 4 skq .., rows srow + 32 v
  int tkq[VB], tnq[VB];                                       // transposing: quad e = tid + 256 v -> (k = 4 tkq .., columns 4 tnq ..), parity of tkq = e & 1
#pragma unroll
  for (int v = 0; v < VB; ++v) { const int e = tid + P4_STAGERS * v, h = e >> 1, hq = h / NQ; tkq[v] = (e & 1) + 2 * hq; tnq[v] = h - hq * NQ; }
  // unconditional, clamped loads (see dppm_wsplit.h): out-of-range threads re-read element 0 and store zeros
  auto load_straight = [&](float4* dst, const float* M, int nr, int k0) {      // M: first row of the block
    const int k = k0 + 4 * skq;
#pragma unroll
    for (int v = 0; v < VA; ++v) {
      const int row = srow + 32 * v;
      dst[v] = *reinterpret_cast<const float4*>(M + (size_t)(row < nr ? row : 0) * ld + (k < ld ? k : 0));
    }
  };
  auto load_transposing = [&](float4* dst, const float* M, int wc, int k0) {   // M: first column of the block
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      const int k = k0 + 4 * tkq[v];
      const float* b0 = M + ((tkq[v] < 8 && 4 * tnq[v] < wc) ? 4 * tnq[v] : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[4 * v + r] = *reinterpret_cast<const float4*>(b0 + (size_t)(k + r < n ? k + r : 0) * ld);
    }
  };
  auto land_straight = [&](const float4* src, char* img, int nr, int k0) {
    const int k = k0 + 4 * skq;
    const bool k0ok = k + 0 < n, k1ok = k + 1 < n, k2ok = k + 2 < n, k3ok = k + 3 < n;
    const unsigned cb = skq >> 1, hb = (skq & 1) * 8;
#pragma unroll
    for (int v = 0; v < VA; ++v) {
      const int row = srow + 32 * v;
      if (row < Cfg::ROWS) {
        const bool r = row < nr;
        const float4 x = src[v];
        p4_put<NS>(img, p4_off(row, cb) + hb, Cfg::IMG, r && k0ok ? x.x : 0.f, r && k1ok ? x.y : 0.f, r && k2ok ? x.z : 0.f, r && k3ok ? x.w : 0.f);
      }
    }
  };
  auto land_transposing = [&](const float4* src, char* img, int nc, int k0) {
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      if (tkq[v] < 8) {
        const int k = k0 + 4 * tkq[v];
        const bool k0ok = k + 0 < n, k1ok = k + 1 < n, k2ok = k + 2 < n, k3ok = k + 3 < n;
        const unsigned cb = tkq[v] >> 1, hb = (tkq[v] & 1) * 8;
        const int row = 4 * tnq[v];
        const float4 x0 = src[4 * v], x1 = src[4 * v + 1], x2 = src[4 * v + 2], x3 = src[4 * v + 3];
        const bool c0 = row + 0 < nc, c1 = row + 1 < nc, c2 = row + 2 < nc, c3 = row + 3 < nc;
        p4_put<NS>(img, p4_off(row + 0, cb) + hb, Cfg::IMG, c0 && k0ok ? x0.x : 0.f, c0 && k1ok ? x1.x : 0.f, c0 && k2ok ? x2.x : 0.f, c0 && k3ok ? x3.x : 0.f);
        p4_put<NS>(img, p4_off(row + 1, cb) + hb, Cfg::IMG, c1 && k0ok ? x0.y : 0.f, c1 && k1ok ? x1.y : 0.f, c1 && k2ok ? x2.y : 0.f, c1 && k3ok ? x3.y : 0.f);
        p4_put<NS>(img, p4_off(row + 2, cb) + hb, Cfg::IMG, c2 && k0ok ? x0.z : 0.f, c2 && k1ok ? x1.z : 0.f, c2 && k2ok ? x2.z : 0.f, c2 && k3ok ? x3.z : 0.f);
        p4_put<NS>(img, p4_off(row + 3, cb) + hb, Cfg::IMG, c3 && k0ok ? x0.w : 0.f, c3 && k1ok ? x1.w : 0.f, c3 && k2ok ? x2.w : 0.f, c3 && k3ok ? x3.w : 0.f);
      }
    }
  };
  constexpr int NX = TA ? 4 * VB : VA, NY = TB ? VA : 4 * VB;
  typedef P4Pre<NX, NY> Pre;
  int ii = 0, ikp = 0;                                        // issue cursor (plane, panel); clamped at the stream's end
  auto issue = [&](Pre& pre) {
    if (dbg & 4) return;
    const long g = (long)wk.group + (long)ii * wk.ngroups;
    const int k0 = ikp * P3_BK;
    const float* A = p.A + g * plane;
    const float* B = p.B + g * plane;
    if (TA) load_transposing(pre.x, A + wk.mlo, wrows, k0); else load_straight(pre.x, A + (size_t)wk.mlo * ld, nrows, k0);
    if (TB) load_straight(pre.y, B + (size_t)wk.nlo * ld, ncols, k0); else load_transposing(pre.y, B + wk.nlo, wcols, k0);
    if (ii < nloc - 1 || ikp < np - 1) { if (++ikp == np) { ikp = 0; ++ii; } }
  };
  int lkp = 0;                                                // landing cursor (panel index inside its plane)
  auto land = [&](const Pre& pre, char* stage) {
    if (dbg & 2) return;
    const int k0 = lkp * P3_BK;
    if (TA) land_transposing(pre.x, stage, nrows, k0); else land_straight(pre.x, stage, nrows, k0);
    if (TB) land_straight(pre.y, stage + OPB, ncols, k0); else land_transposing(pre.y, stage + OPB, ncols, k0);
    if (++lkp == np) lkp = 0;
  };

  Pre s0, s1;
  issue(s0); issue(s1);                                       // panels 0, 1
  land(s0, lds4);
  issue(s0);                                                  // panel 2
  __syncthreads();                                            // panel 0 is in stage 0
  // interval q: the multipliers work on panel q; panel q+1 lands in the other stage from set (q+1) & 1, which then takes panel q+3
  for (int q = 0; q < nq; q += 2) {
    if (q + 1 < nq) { land(s1, lds4 + STAGE); issue(s1); }
    __syncthreads();
    if (q + 1 < nq) {
      if (q + 2 < nq) { land(s0, lds4); issue(s0); }
      __syncthreads();
    }
  }
}

static int p4_dbg() { return sw(SW_DBG_P4); }   // timing experiments only: results are wrong

template <int NT, int NS, int NBLK, bool TA, bool TB>
static int plane4_launch_nt(const PlaneArgs& a, int batch, hipStream_t st) {
  const size_t lds = (size_t)2 * 2 * P4Img<NS>::N * P4Cfg<NT>::IMG;   // stages x operands x images
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(&plane4_kernel<NT, NS, NBLK, TA, TB>), (size_t)(lds), "plane4"));
  // one workgroup per CU; NBLK = 1: planes dealt round-robin; NBLK = 2: groups of four workgroups (see P4Work), all 256 launched
  const int grid = NBLK == 1 ? (batch < P4_CUS ? batch : P4_CUS) : P4_CUS;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(plane4_kernel<NT, NS, NBLK, TA, TB>), dim3(grid), dim3(P4_THREADS), lds, st, a, batch, p4_dbg());
  return check_launch("plane4");
}

// ns = 1, 3 (inference, forward form): 129 <= n <= 224 one block per plane; 225 <= n <= 512 2 x 2 blocks (enough planes to fill the
// 64 groups).  ns = 6 (float32-equivalent, all three forms): 129 <= n <= 208.
bool plane4_ok(const PlaneArgs& a, int batch, int ns, bool ta, bool tb) {
  if (a.ld % 4 != 0 || a.ld < a.n || batch <= 0 || (long)batch * a.n * a.ld >= (1L << 40) || (ta && tb)) return false;
  const int nt = cdiv(a.n, 16);
  if (ns == 6) return nt >= 9 && nt <= 13;
  if (ta || tb) return false;
  return (nt >= 9 && nt <= 14) || (nt >= 15 && nt <= 32 && batch >= P4_CUS / 8);
}

template <int NS>
static int plane4_dispatch(const PlaneArgs& a, int batch, hipStream_t st) {
  const int nt = cdiv(a.n, 16);
  if (nt <= 14) switch (nt) {
    case 9: return plane4_launch_nt<9, NS, 1, false, false>(a, batch, st);
    case 10: return plane4_launch_nt<10, NS, 1, false, false>(a, batch, st);
    case 11: return plane4_launch_nt<11, NS, 1, false, false>(a, batch, st);
    case 12: return plane4_launch_nt<12, NS, 1, false, false>(a, batch, st);
    case 13: return plane4_launch_nt<13, NS, 1, false, false>(a, batch, st);
    case 14: return plane4_launch_nt<14, NS, 1, false, false>(a, batch, st);
    default: break;
  } else switch (cdiv(nt, 2)) {
    case 8: return plane4_launch_nt<8, NS, 2, false, false>(a, batch, st);
    case 9: return plane4_launch_nt<9, NS, 2, false, false>(a, batch, st);
    case 10: return plane4_launch_nt<10, NS, 2, false, false>(a, batch, st);
    case 11: return plane4_launch_nt<11, NS, 2, false, false>(a, batch, st);
    case 12: return plane4_launch_nt<12, NS, 2, false, false>(a, batch, st);
    case 13: return plane4_launch_nt<13, NS, 2, false, false>(a, batch, st);
    case 14: return plane4_launch_nt<14, NS, 2, false, false>(a, batch, st);
    case 15: return plane4_launch_nt<15, NS, 2, false, false>(a, batch, st);
    case 16: return plane4_launch_nt<16, NS, 2, false, false>(a, batch, st);
    default: break;
  }
  return fail(DPPM_E_UNSUPPORTED, "plane4: n=%d outside 129..512", a.n);
}

template <bool TA, bool TB>
static int plane6_dispatch(const PlaneArgs& a, int batch, hipStream_t st) {
  switch (cdiv(a.n, 16)) {
    case 9: return plane4_launch_nt<9, 6, 1, TA, TB>(a, batch, st);
    case 10: return plane4_launch_nt<10, 6, 1, TA, TB>(a, batch, st);
    case 11: return plane4_launch_nt<11, 6, 1, TA, TB>(a, batch, st);
    case 12: return plane4_launch_nt<12, 6, 1, TA, TB>(a, batch, st);
    case 13: return plane4_launch_nt<13, 6, 1, TA, TB>(a, batch, st);
    default: return fail(DPPM_E_UNSUPPORTED, "plane4: n=%d outside 129..208 for the three-piece split", a.n);
  }
}

int plane4_run(const PlaneArgs& a, int batch, int ns, bool ta, bool tb, hipStream_t st) {
  if (!plane4_ok(a, batch, ns, ta, tb)) return fail(DPPM_E_UNSUPPORTED, "plane4: n=%d ns=%d ta=%d tb=%d batch=%d not supported", a.n, ns, (int)ta, (int)tb, batch);
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  int rc;
  if (ns == 6) rc = ta ? plane6_dispatch<true, false>(a, batch, st) : tb ? plane6_dispatch<false, true>(a, batch, st) : plane6_dispatch<false, false>(a, batch, st);
  else rc = ns > 1 ? plane4_dispatch<3>(a, batch, st) : plane4_dispatch<1>(a, batch, st);
  if (prof) prof_after(st, PROF_PLANE_MM, prof_arith_of_pieces(ns), 2.0 * a.n * a.n * (double)a.n * batch, 3.0 * sizeof(float) * a.n * (double)a.ld * batch);
  return rc;
}

}  // namespace dppm
