// Role-split forms of the split-bf16 H = 128 linears (see dppm_w4.h): staging waves and multiplying waves.
//
// The kernels of dppm_wsplit.h let every wave both stage tiles (global -> registers -> bf16 hi / lo images in LDS) and multiply
// them: at 128 registers per wave that leaves room for one tile in flight per thread, and every wait for it sits on the path of
// the MFMAs.  Here waves 0-7 only stage -- three tiles in flight per thread, plus the (scale, shift) pair of the fused instance
// norm, no accumulators, no weights -- and waves 8-15 only multiply: both weight matrices' fragments stay in 64 registers, the
// X fragments come from LDS, results go out with 16-byte stores.  One barrier per staged tile keeps the two groups one tile
// apart; the stagers' loads never wait for a multiplier and the multipliers never wait for HBM.
#include "dppm_w4.h"
#include "dppm_wsplit.h"

namespace dppm {

constexpr int W4_THREADS = 1024;
#ifndef W4_HALF
#define W4_HALF 0
#endif
// Timing ablations, compile time only (-DW4_ABLATE=bits; results are wrong on purpose): 1 no output stores, 2 no second-product MFMAs,
// 4 no first-product MFMAs, 8 the stagers request no new tiles (tools/dbg/mlp_ablate.sh)
#ifndef W4_ABLATE
#define W4_ABLATE 0
#endif
// Timing instrument, compiled in with -DW4_TRACE_ON=1 only: with DPPM_DBG_W4=2048 workgroup 8 of w4_mlp_kernel records, per barrier
// interval, the 100 MHz wall clock at which one wave of each role (0 stager, 1 first product, 2 second product) leaves the barrier
// (k = 0), reaches the next one (k = 1) and ends its MFMA loop (k = 2); the 21st launch writes gpurun_out/w4_trace.txt.
#ifndef W4_TRACE_ON
#define W4_TRACE_ON 0
#endif
#if W4_TRACE_ON
__device__ unsigned long long g_w4_trace[3 * 3 * 256];
#define W4_TRACE(role, i, k) do { if ((a.dbg & 2048) && blockIdx.x == 8 && (i) < 256 && (tid & 63) == 0) g_w4_trace[((i) * 3 + (role)) * 3 + (k)] = wall_clock64(); } while (0)
#else
#define W4_TRACE(role, i, k) do { } while (0)
#endif

// (stager thread map w4_q / w4_cg: dppm_wsplit.h)
#ifndef W4_PKEPI
#define W4_PKEPI 0
#endif
#ifndef W4_SKEW
#define W4_SKEW 0
#endif
#ifndef W4_BIASINIT
#define W4_BIASINIT 0   // experiment, off: both products' accumulators start from the bias (64 of ~550 vector instructions per tile and SIMD less):
                        // 0.4868 -> 0.4817 ms, and the headline distance to float64 goes from 1.01e-4 to 1.30e-4 -- not worth a new rounding order
#endif
__device__ __forceinline__ floatx2 w4_pk_add(floatx2 a, floatx2 b) { floatx2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ floatx2 w4_pk_mul(floatx2 a, floatx2 b) { floatx2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
struct W4Pre { float4 a, b, c, d, sc, sh; float m; };   // one tile in flight per staging thread (named members: registers)

// w3_mma4 with the LDS fragment reads one (k step, position group) ahead of the MFMAs that use them: a multiplying wave shares its
// SIMD with one other multiplier only, so the read latency has to be covered inside the wave (8 more registers).
template <int NS, bool W_IS_A>
__device__ __forceinline__ void w4_mma4(floatx4 (&acc)[4], const uintx4 (&wh)[4], const uintx4 (&wl)[4], const char* __restrict__ T, const W3Addr& ad) {
  uintx4 xh = *reinterpret_cast<const uintx4*>(T + w3_off(ad, 0, 0)), xl = xh;
  if (NS > 1) xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + w3_off(ad, 0, 0));
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int s = n >> 2, j = n & 3;
    const uintx4 ch_ = xh, cl_ = xl;
    if (n + 1 < 16) {
      const unsigned off = w3_off(ad, (n + 1) & 3, (n + 1) >> 2);
      xh = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) xl = *reinterpret_cast<const uintx4*>(T + W3_IMG + off);
    }
    if (NS > 1) {
      acc[j] = W_IS_A ? w3_mfma(wh[s], cl_, acc[j]) : w3_mfma(cl_, wh[s], acc[j]);
      acc[j] = W_IS_A ? w3_mfma(wl[s], ch_, acc[j]) : w3_mfma(ch_, wl[s], acc[j]);
    }
    acc[j] = W_IS_A ? w3_mfma(wh[s], ch_, acc[j]) : w3_mfma(ch_, wh[s], acc[j]);
  }
}

// The same for a wave that owns two 16-channel blocks: every X fragment feeds both (six MFMAs per pair of LDS reads).  The order
// reads-then-MFMAs is pinned, or the scheduler hoists a tile's worth of reads over the accumulators' registers.
template <int NS, bool W_IS_A>
__device__ __forceinline__ void w4_mma8(floatx4 (&acc)[2][4], const uintx4 (&wh)[2][4], const uintx4 (&wl)[2][4], const char* __restrict__ T, const W3Addr& ad) {
  // Two fragment sets in turn (no copies): while the six MFMAs of step n run from one set, the reads of step n+1 are already in
  // flight into the other.  sched_barrier(0) keeps that program order: with sched_group_barrier alone the scheduler paired every
  // read group with the MFMAs that CONSUME it (reads, wait, MFMAs -- the LDS latency in front of every step).
  uintx4 xh[2], xl[2];
  xh[0] = *reinterpret_cast<const uintx4*>(T + w3_off(ad, 0, 0));
  xl[0] = xh[0];
  if (NS > 1) xl[0] = *reinterpret_cast<const uintx4*>(T + W3_IMG + w3_off(ad, 0, 0));
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int s = n >> 2, j = n & 3, cur = n & 1, nxt = cur ^ 1;
    if (n + 1 < 16) {
      const unsigned off = w3_off(ad, (n + 1) & 3, (n + 1) >> 2);
      xh[nxt] = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) xl[nxt] = *reinterpret_cast<const uintx4*>(T + W3_IMG + off); else xl[nxt] = xh[nxt];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      if (NS > 1) {
        acc[cb][j] = W_IS_A ? w3_mfma(wh[cb][s], xl[cur], acc[cb][j]) : w3_mfma(xl[cur], wh[cb][s], acc[cb][j]);
        acc[cb][j] = W_IS_A ? w3_mfma(wl[cb][s], xh[cur], acc[cb][j]) : w3_mfma(xh[cur], wl[cb][s], acc[cb][j]);
      }
      acc[cb][j] = W_IS_A ? w3_mfma(wh[cb][s], xh[cur], acc[cb][j]) : w3_mfma(xh[cur], wh[cb][s], acc[cb][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// w4_mma8 over the position groups J0 .. J0 + NJ - 1 only, with a hook after every k step: `between(n)` is program text that the
// scheduler may not move (sched_barrier on both sides), i.e. vector work of ANOTHER half tile issued in the shadow of these MFMAs.
template <int NS, bool W_IS_A, int J0, int NJ, typename F>
__device__ __forceinline__ void w4_mma8_part(floatx4 (&acc)[2][4], const uintx4 (&wh)[2][4], const uintx4 (&wl)[2][4], const char* __restrict__ T, const W3Addr& ad,
                                             F&& between) {
  uintx4 xh[2], xl[2];
  xh[0] = *reinterpret_cast<const uintx4*>(T + w3_off(ad, J0, 0));
  xl[0] = xh[0];
  if (NS > 1) xl[0] = *reinterpret_cast<const uintx4*>(T + W3_IMG + w3_off(ad, J0, 0));
#pragma unroll
  for (int n = 0; n < 4 * NJ; ++n) {
    const int s = n / NJ, j = J0 + n % NJ, cur = n & 1, nxt = cur ^ 1;
    if (n + 1 < 4 * NJ) {
      const unsigned off = w3_off(ad, J0 + (n + 1) % NJ, (n + 1) / NJ);
      xh[nxt] = *reinterpret_cast<const uintx4*>(T + off);
      if (NS > 1) xl[nxt] = *reinterpret_cast<const uintx4*>(T + W3_IMG + off); else xl[nxt] = xh[nxt];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      if (NS > 1) {
        acc[cb][j] = W_IS_A ? w3_mfma(wh[cb][s], xl[cur], acc[cb][j]) : w3_mfma(xl[cur], wh[cb][s], acc[cb][j]);
        acc[cb][j] = W_IS_A ? w3_mfma(wl[cb][s], xh[cur], acc[cb][j]) : w3_mfma(xh[cur], wl[cb][s], acc[cb][j]);
      }
      acc[cb][j] = W_IS_A ? w3_mfma(wh[cb][s], xh[cur], acc[cb][j]) : w3_mfma(xh[cur], wh[cb][s], acc[cb][j]);
    }
    between(n);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// in_lin while staging (WsArgs::inl_*): the staging thread's four channels 4 cg .. 4 cg + 3 at its four positions from the two
// input values of each position
struct W4Inl { float w0[4], w1[4], b[4]; };
__device__ __forceinline__ W4Inl w4_inl_load(const WsArgs& a, int cg) {
  W4Inl r;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int c = 4 * cg + v;
    r.w0[v] = a.inl_w[2 * c]; r.w1[v] = a.inl_w[2 * c + 1]; r.b[v] = a.inl_b ? a.inl_b[c] : 0.f;
  }
  return r;
}
// ws_inlin's three roundings as single-lane instructions that the compiler cannot pack.  Left to itself hipcc turns the rows below into
// v_pk_mul_f32 / v_pk_fma_f32 with operand-half selects (op_sel) applied straight to the destination pairs of the stagers' global
// loads, and every build of that shape gave, from run to run, wrong values at the odd positions of a thread's quad in tiles after a
// workgroup's first; every build without those packed forms is bit-exact ("The in_lin stagers": evidence,
// what was ruled out, the two hardware probes).  -DW4_INLIN_PACKED restores the packed form for that experiment.
__device__ __forceinline__ float w4_inlin_s(float w0, float w1, float bc, float a, float s) {
#ifdef W4_INLIN_PACKED
  return ws_inlin(w0, w1, bc, a, s);
#else
  return w3_side_s(w0, w1, bc, a, s);
#endif
}
__device__ __forceinline__ float4 w4_inl_row(const W4Inl& k, int v, const float4& A, const float4& S) {
  return make_float4(w4_inlin_s(k.w0[v], k.w1[v], k.b[v], A.x, S.x), w4_inlin_s(k.w0[v], k.w1[v], k.b[v], A.y, S.y),
                     w4_inlin_s(k.w0[v], k.w1[v], k.b[v], A.z, S.z), w4_inlin_s(k.w0[v], k.w1[v], k.b[v], A.w, S.w));
}

// Y = W1 X1 + W2 norm(X2) + b per 64-position tile (model/ppgn.py:66-69: m4 on cat(mm, x)); NRM: X2 is staged through its instance
// norm (a.aff) and the result's masked per-tile statistics are written (a.stat, a.cnt), see WsArgs in dppm_wstat.h.
// X tiles in stream order: q = 2 i + s (tile i of this workgroup, source s); tile q travels in register set q % 3 and lands in
// LDS buffer q & 1.
template <int NS, bool NRM, bool INLIN>
__global__ __launch_bounds__(W4_THREADS) void w4_lin2_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  char* Xs = smem4;                                          // [2][BUF]
  float* Ms = reinterpret_cast<float*>(smem4 + 2 * BUF);     // [2][64] mask rows of position tiles i, i + 1 (NRM)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 3, w8 = wave & 7;
  const int g = lane >> 4, ci = lane & 15, ch = 16 * w8 + ci;
  const WsJob& J = a.job[0];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  const int rank = blockIdx.x, G = gridDim.x;
  if (rank >= ntiles) return;                                // whole workgroup, before any barrier
  const int nloc = (ntiles - rank + G - 1) / G;              // position tiles of this workgroup: rank + i * G
  const int nq = 2 * nloc;

  if (role == 0) {
    // ------------------------------------------------------------------------------------------------ stagers
    const unsigned sq = w4_q(tid), scg = w4_cg(tid);
    const unsigned o0 = (4 * scg) * P + 4 * sq;              // channels 4 cg .., positions 4 q ..
    const int oc = 4 * sq;
    int qi = 0, qs = 0;                                      // issue cursor (tile, source)
    W4Inl inl{};
    if (INLIN) inl = w4_inl_load(a, scg);
    auto issue = [&](W4Pre& pre) {                           // unconditional clamped loads of the tile at the cursor
      const int t = rank + (qi < nloc ? qi : nloc - 1) * G;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
      const float* Xt;
      size_t s1 = P, s2 = (size_t)2 * P, s3 = (size_t)3 * P;
      if (INLIN) {   // source 1 = the two input planes (a: A, b: sigma I; c, d re-read them): pointer and stride selects, no branch
        Xt = qs ? a.inl_in + (size_t)b * a.inl_bs + p0 + ((p0 + oc < P) ? (unsigned)oc : 0u) : J.X + (size_t)b * J.x_bs + p0 + ((p0 + oc < P) ? o0 : 0u);
        s2 = qs ? 0 : s2; s3 = qs ? s1 : s3;
      } else {
        Xt = (qs ? J.X2 + (size_t)b * J.x2_bs : J.X + (size_t)b * J.x_bs) + p0 + ((p0 + oc < P) ? o0 : 0u);
      }
      pre.a = *reinterpret_cast<const float4*>(Xt);
      pre.b = *reinterpret_cast<const float4*>(Xt + s1);
      pre.c = *reinterpret_cast<const float4*>(Xt + s2);
      pre.d = *reinterpret_cast<const float4*>(Xt + s3);
      if (NRM) {
        const float* T = a.aff + (size_t)b * 256 + 4 * scg;
        pre.sc = *reinterpret_cast<const float4*>(T);
        pre.sh = *reinterpret_cast<const float4*>(T + 128);
        pre.m = a.maskstat[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
      }
      if (++qs == 2) { qs = 0; ++qi; }
    };
    int li = 0, ls = 0;                                      // landing cursor
    auto land = [&](const W4Pre& pre, int buf) {
      const int t = rank + (li < nloc ? li : nloc - 1) * G;
      const int b = t / tpp, p0 = (t - b * tpp) * WS_TP;
      const bool in = li < nloc && p0 + oc < P;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xa = in ? pre.a : z, xb = in ? pre.b : z, xc = in ? pre.c : z, xd = in ? pre.d : z;
      if (INLIN && ls == 1) {                                // the skip input is u0 = in_lin(A, sigma I), formed here (uniform branch)
        const float4 A4 = xa, S4 = xb;
        xa = w4_inl_row(inl, 0, A4, S4); xb = w4_inl_row(inl, 1, A4, S4); xc = w4_inl_row(inl, 2, A4, S4); xd = w4_inl_row(inl, 3, A4, S4);
        if (!in) xa = xb = xc = xd = z;
      } else if (NRM && ls == 1) {                           // the skip input's instance norm (uniform branch)
        w3_affine<true>(xa, pre.sc.x, pre.sh.x); w3_affine<true>(xb, pre.sc.y, pre.sh.y); w3_affine<true>(xc, pre.sc.z, pre.sh.z); w3_affine<true>(xd, pre.sc.w, pre.sh.w);
      }
      w3_tile_store4_qc<NS>(xa, xb, xc, xd, Xs + buf * BUF, sq, scg);
      if (NRM && ls == 0 && tid < 64) Ms[(li & 1) * 64 + tid] = (li < nloc && p0 + lane < P) ? pre.m : 0.f;
      if (++ls == 2) { ls = 0; ++li; }
    };
    W4Pre p0s, p1s, p2s;
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
  uintx4 w1h[4], w1l[4], w2h[4], w2l[4];
  w3_load_w<NS, false>(w1h, w1l, J.W1, J.ldw1, ch, g);
  w3_load_w<NS, false>(w2h, w2l, J.W2, J.ldw2, ch, g);
  const float bv = J.b1 ? J.b1[ch] : 0.f;
  const W3Addr ad = w3_addr(ci, g);
  const unsigned r0 = (unsigned)ch * P + 4 * g;
  const unsigned m0 = 4 * g;
  __syncthreads();                                           // tile 0 is in buffer 0
  for (int i = 0; i < nloc; ++i) {
    const int t = rank + i * G;
    const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    w4_mma4<NS, false>(acc, w1h, w1l, Xs, ad);
    __syncthreads();
    w4_mma4<NS, false>(acc, w2h, w2l, Xs + BUF, ad);
    if (NRM) {
      // masked statistics of the result over the tile (one pass about a pivot inside the data, see w3_lin2_kernel)
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
      const float dm = s1 * __builtin_amdgcn_rcpf(fmaxf(n, 1.f));
      if (g == 0) *reinterpret_cast<float2*>(a.stat + ((size_t)t * WS_H + ch) * 2) = make_float2(piv + dm, fmaxf(s2 - s1 * dm, 0.f));
      if (w8 == 0 && lane == 0) a.cnt[t] = n;
    }
    // NRM: the RAW result is what the layer tensor holds from here on (its norm is applied by the consumers), with masked positions
    // zeroed as the reference's norm does (model/ppgn.py:315): they restart from `shift` in every consumer and cannot grow from layer
    // to layer (ws_mask_mul: 0 * anything = 0).  Unmasked results do not depend on them (a position is a column of every product).
    ws_store_out(acc, J.Y + (size_t)b * J.y_bs + p0, r0, NRM ? Ms + (i & 1) * 64 : nullptr, m0, P - p0, bv, false, false);   // (wide stores, ws_store_wide / _lds: 0.41 - 0.42 ms against 0.40 ms)
    __syncthreads();
  }
}

// Fused MLP  out = (W2 LeakyReLU(W1 norm(X) + b1) + b2) * mask  (model/ppgn.py:58-59), one job (m1 or m2) per workgroup.
// Stagers (waves 0 .. STG-1): X tiles (NRM: through the source's instance norm, a.aff) and the tile's mask row.  Per barrier
// interval the next four waves start tile i (first product from X[i & 1], bias, LeakyReLU, split into the hidden image H[i & 1])
// while the last four finish tile i-1 (second product from H[(i-1) & 1], bias, mask, 16-byte stores): one barrier per tile, X and
// H double-buffered.  Tile i travels in the stagers' register set i % 3.
// STG = 4 (the form launched): 12 waves at 168 registers -- every staging thread carries two channel groups (4 cg .. and
// 64 + 4 cg ..) and the multipliers have room for a second X fragment set (w4_mma8's reads one step ahead).  STG = 8 would be 16
// waves at 128 registers, one channel group per staging thread.
// NRM: the graph's 256-float (scale, shift) table goes through LDS instead of 16 registers per tile in flight: with tile q the
// stagers request one float4 of the table of tile q+1's graph (pre.t), write it to As[(q+1) & 1] when tile q lands, and read their
// channels' pairs from As[q' & 1] when tile q' = q+1 lands one barrier later.
template <int NQ> struct W4PreT;
template <> struct W4PreT<1> { float4 a, b, c, d, t; float m; };
template <> struct W4PreT<2> { float4 a, b, c, d, e, f, g, h, t; float m; };
template <int NS, bool NRM, bool INLIN, int STG>
__global__ __launch_bounds__(64 * (STG + 8)) void w4_mlp_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  constexpr int NIMG = NS > 1 ? 2 : 1;
  constexpr int BUF = NIMG * W3_IMG;
  constexpr int NQ = 8 / STG;                                // channel groups per staging thread
  char* Xs = smem4;                                          // [2][BUF]
  char* Hs = smem4 + 2 * BUF;                                // [2][BUF]
  float* Ms = reinterpret_cast<float*>(smem4 + 4 * BUF);     // [4][64] mask rows: tile i is written in step i-1 and read in step i+1
  float* As = Ms + 4 * 64;                                   // [2][256] (scale, shift) tables of the graphs of tiles i, i+1 (NRM)
  float* Rs = As + 2 * 256;                                  // [4][16][WS_WIDE_PITCH] the second product's waves: result blocks on their way out
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ci = lane & 15;
  const WsWho who = ws_who(a.njobs);
  const WsJob& J = a.job[who.jid];
  const int P = a.P, tpp = (P + WS_TP - 1) / WS_TP, ntiles = a.B * tpp;
  if (who.rank >= ntiles) return;                            // whole workgroup, before any barrier
  const int nloc = (ntiles - who.rank + who.per_job - 1) / who.per_job;   // tiles rank + i * per_job
  const float* maskp = a.maskp;

  if (wave < STG) {
    // ------------------------------------------------------------------------------------------------ stagers
    using Pre = W4PreT<NQ>;
    const unsigned sq = w4_q(tid), scg = w4_cg(tid);
    const unsigned o0 = (4 * scg) * P + 4 * sq;
    const int oc = 4 * sq;
    int qi = 0;
    W4Inl inl{}, inl2{};
    if (INLIN) { inl = w4_inl_load(a, scg); if (NQ == 2) inl2 = w4_inl_load(a, scg + 16); }
    auto issue = [&](Pre& pre) {
      const int t = who.rank + (qi < nloc ? qi : nloc - 1) * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
      if (INLIN) {   // the source is u0 = in_lin(A, sigma I): two input planes instead of 128 channels
        const float* It = a.inl_in + (size_t)b * a.inl_bs + p0 + ((p0 + oc < P) ? (unsigned)oc : 0u);
        pre.a = *reinterpret_cast<const float4*>(It);
        pre.b = *reinterpret_cast<const float4*>(It + (size_t)P);
      } else {
        const float* Xt = J.X + (size_t)b * J.x_bs + p0 + ((p0 + oc < P) ? o0 : 0u);
        pre.a = *reinterpret_cast<const float4*>(Xt);
        pre.b = *reinterpret_cast<const float4*>(Xt + (size_t)P);
        pre.c = *reinterpret_cast<const float4*>(Xt + (size_t)2 * P);
        pre.d = *reinterpret_cast<const float4*>(Xt + (size_t)3 * P);
        if constexpr (NQ == 2) {
          const float* Xu = Xt + (size_t)64 * P;
          pre.e = *reinterpret_cast<const float4*>(Xu);
          pre.f = *reinterpret_cast<const float4*>(Xu + (size_t)P);
          pre.g = *reinterpret_cast<const float4*>(Xu + (size_t)2 * P);
          pre.h = *reinterpret_cast<const float4*>(Xu + (size_t)3 * P);
        }
      }
      if (NRM) {
        const int t1 = who.rank + (qi + 1 < nloc ? qi + 1 : nloc - 1) * who.per_job;
        pre.t = *reinterpret_cast<const float4*>(a.aff + (size_t)__builtin_amdgcn_readfirstlane(t1 / tpp) * 256 + 4 * lane);
      }
      if (maskp) pre.m = maskp[(size_t)b * P + ((p0 + lane < P) ? p0 + lane : 0)];
      ++qi;
    };
    int li = 0;
    auto land = [&](const Pre& pre) {                        // tile li -> X[li & 1], mask row li & 3
      const int t = who.rank + (li < nloc ? li : nloc - 1) * who.per_job;
      const int b = t / tpp, p0 = (t - b * tpp) * WS_TP;
      // No zero fill: positions past the plane's end and landings past the last tile hold clamped duplicates that no store uses.
      char* Xb = Xs + (li & 1) * BUF;
      const float* Ab = As + (li & 1) * 256 + 4 * scg;
      {
        float4 xa, xb, xc, xd;
        if (INLIN) {
          xa = w4_inl_row(inl, 0, pre.a, pre.b); xb = w4_inl_row(inl, 1, pre.a, pre.b); xc = w4_inl_row(inl, 2, pre.a, pre.b); xd = w4_inl_row(inl, 3, pre.a, pre.b);
        } else {
          xa = pre.a; xb = pre.b; xc = pre.c; xd = pre.d;
        }
        if (NRM) {
          const float4 sc = *reinterpret_cast<const float4*>(Ab), sh = *reinterpret_cast<const float4*>(Ab + 128);
          w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
        }
        w3_tile_store4_qc<NS>(xa, xb, xc, xd, Xb, sq, scg);
      }
      if constexpr (NQ == 2) {                               // channels 64 + 4 cg ..: channel group cg + 16
        float4 xa, xb, xc, xd;
        if constexpr (INLIN) {
          xa = w4_inl_row(inl2, 0, pre.a, pre.b); xb = w4_inl_row(inl2, 1, pre.a, pre.b); xc = w4_inl_row(inl2, 2, pre.a, pre.b); xd = w4_inl_row(inl2, 3, pre.a, pre.b);
        } else {
          xa = pre.e; xb = pre.f; xc = pre.g; xd = pre.h;
        }
        if (NRM) {
          const float4 sc = *reinterpret_cast<const float4*>(Ab + 64), sh = *reinterpret_cast<const float4*>(Ab + 192);
          w3_affine(xa, sc.x, sh.x); w3_affine(xb, sc.y, sh.y); w3_affine(xc, sc.z, sh.z); w3_affine(xd, sc.w, sh.w);
        }
        w3_tile_store4_qc<NS>(xa, xb, xc, xd, Xb, sq, scg + 16);
      }
      if (tid < 64) {
        Ms[(li & 3) * 64 + tid] = (maskp && li < nloc && p0 + lane < P) ? pre.m : (maskp ? 0.f : 1.f);
        if (NRM) *reinterpret_cast<float4*>(As + ((li + 1) & 1) * 256 + 4 * tid) = pre.t;
      }
      ++li;
    };
    Pre p0s, p1s, p2s;
    p0s.m = p1s.m = p2s.m = 1.f;
    if (NRM) {                                               // tile 0's table, directly
      if (tid < 64) *reinterpret_cast<float4*>(As + 4 * tid) = *reinterpret_cast<const float4*>(a.aff + (size_t)(who.rank / tpp) * 256 + 4 * tid);
      __syncthreads();
    }
    issue(p0s); issue(p1s); issue(p2s);
    land(p0s);
    __syncthreads();                                         // tile 0 is in X[0]
    // steps i = 0 .. nloc: tile i's first product runs, tile i+1 lands, tile i+3 is requested
    for (int q = 0; q <= nloc; q += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (q + u <= nloc) {
          if (wave == 0) W4_TRACE(0, q + u, 0);
          if (W4_ABLATE & 8) { if (u == 0) land(p1s); else if (u == 1) land(p2s); else land(p0s); } else
          if (u == 0) { issue(p0s); land(p1s); }
          else if (u == 1) { issue(p1s); land(p2s); }
          else { issue(p2s); land(p0s); }
          if (wave == 0) W4_TRACE(0, q + u, 1);
          __syncthreads();
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- multipliers
  // The first four take the first product, the last four the second; each owns 32 channels (two 16-channel blocks), so that one X
  // fragment read from LDS feeds six MFMAs instead of three: with 16 channels per wave the matrix pipe (45 % busy) and the LDS
  // (47 %) took turns inside every wave (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES / SQ_LDS_IDX_ACTIVE of w3_mlp_kernel).
  const int second = ((wave - STG) >> 2) & 1, pw = (wave - STG) & 3;   // product, channel group 32 pw ..
  uintx4 wh[2][4], wl[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
    w3_load_w<NS, false>(wh[cb], wl[cb], second ? J.W2 : J.W1, second ? J.ldw2 : J.ldw1, 32 * pw + 16 * cb + ci, g);
  const W3Addr ad = w3_addr(ci, g);
  if (NRM) __syncthreads();                                  // (the stagers' table of tile 0)
  __syncthreads();                                           // tile 0 is in X[0]
  if (!second) {
    // first product, D = W1 X: the lane holds channels 32 pw + 16 cb + 4 g + r of position 16 j + ci -> hidden image
    const unsigned hbase = ci * 256 + (g & 1) * 8;
#if W4_PKEPI
    const floatx2 slope2 = {kSlope, kSlope};
#endif
    float4 b1w[2];                                           // loaded once: inside the loop every tile would wait for a global load
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float* bp = J.b1 + 32 * pw + 16 * cb + 4 * g;
      b1w[cb] = J.b1 ? make_float4(bp[0], bp[1], bp[2], bp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = 0; i <= nloc; ++i) {
      if (pw == 0) W4_TRACE(1, i, 0);
      if (i < nloc) {
        char* Hb = Hs + (i & 1) * BUF;
        floatx4 acc[2][4];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[cb][j] = W4_BIASINIT ? floatx4{b1w[cb].x, b1w[cb].y, b1w[cb].z, b1w[cb].w} : floatx4{0.f, 0.f, 0.f, 0.f};
        // bias, LeakyReLU, split and the hidden-image stores of one (channel block, position group) unit
        auto h_unit = [&](int cb, int j) {
          const float4 b1 = b1w[cb]; (void)b1;
          const unsigned hchunk = (unsigned)(2 * (2 * pw + cb) + (g >> 1)) ^ (unsigned)ci;
          unsigned h01, l01 = 0, h23, l23 = 0;
#if W4_PKEPI
          // (experiment, off: bias and the LeakyReLU's slope product as packed instructions on the accumulator's aligned register pairs;
          // 32 vector instructions fewer per tile and wave, bit-identical, and no faster: 0.5025 against 0.4937 ms together with W3_NATPAIR)
          const floatx2 v01 = w4_pk_add(floatx2{acc[cb][j][0], acc[cb][j][1]}, floatx2{b1.x, b1.y});
          const floatx2 v23 = w4_pk_add(floatx2{acc[cb][j][2], acc[cb][j][3]}, floatx2{b1.z, b1.w});
          const floatx2 s01 = w4_pk_mul(v01, slope2), s23 = w4_pk_mul(v23, slope2);
          w3_split2<NS>(fmaxf(v01[0], s01[0]), fmaxf(v01[1], s01[1]), h01, l01);
          w3_split2<NS>(fmaxf(v23[0], s23[0]), fmaxf(v23[1], s23[1]), h23, l23);
#else
#if W4_BIASINIT
          w3_split2<NS>(ws_leaky(acc[cb][j][0]), ws_leaky(acc[cb][j][1]), h01, l01);
          w3_split2<NS>(ws_leaky(acc[cb][j][2]), ws_leaky(acc[cb][j][3]), h23, l23);
#else
          w3_split2<NS>(ws_leaky(acc[cb][j][0] + b1.x), ws_leaky(acc[cb][j][1] + b1.y), h01, l01);
          w3_split2<NS>(ws_leaky(acc[cb][j][2] + b1.z), ws_leaky(acc[cb][j][3] + b1.w), h23, l23);
#endif
#endif
          const unsigned off = hbase + j * 4096 + ((hchunk ^ j) << 4);
          *reinterpret_cast<uint2*>(Hb + off) = make_uint2(h01, h23);
          if (NS > 1) *reinterpret_cast<uint2*>(Hb + W3_IMG + off) = make_uint2(l01, l23);
        };
#if W4_HALF
        // Two half tiles: the vector work of positions 0 .. 31 is issued between the MFMAs of positions 32 .. 63 (one unit per two k
        // steps), so that only the second half's epilogue is exposed behind the wave's last MFMA.
        const char* Xb = Xs + (i & 1) * BUF;
        if (!(W4_ABLATE & 4))
        {
        w4_mma8_part<NS, true, 0, 2>(acc, wh, wl, Xb, ad, [](int) {});
        w4_mma8_part<NS, true, 2, 2>(acc, wh, wl, Xb, ad, [&](int n) { if (n & 1) h_unit((n >> 1) & 1, n >> 2); });
        }
        if (pw == 0) W4_TRACE(1, i, 2);
#pragma unroll
        for (int j = 2; j < 4; ++j) { h_unit(0, j); h_unit(1, j); }
#else
        w4_mma8<NS, true>(acc, wh, wl, Xs + (i & 1) * BUF, ad);
        if (pw == 0) W4_TRACE(1, i, 2);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) h_unit(cb, j);
#endif
      }
      if (pw == 0) W4_TRACE(1, i, 1);
      __syncthreads();
    }
    return;
  }
  // second product, D = H^T W2^T: the lane holds positions 16 j + 4 g + r of channel 32 pw + 16 cb + ci -> bias, through the wave's
  // LDS patch into 256-byte row pieces, mask, 16-byte stores (ws_store_wide_lds: 0.573 -> 0.538 ms per pair against the result
  // layout's 64-byte pieces; the DPP form of the transposition 0.546 ms).
  // W4_TRACE_ON timeline of one 3.2 us interval (N = 200, B = 32, shader clock 1.88 GHz): first product MFMAs until 1.4 us, hidden
  // image written by 2.7 us; second product MFMAs until 2.1 us (both groups' 192 MFMAs per SIMD are 1.6 us of pipe time), stores out
  // by 3.0 us; stagers done at 2.6 us.  The groups wait for each other's vector and LDS instructions more than for the matrix pipe.
  // Skewing this group against the first product's waves on the same SIMDs -- the stores of tile i-2 at the START of interval i,
  // accumulators kept over the barrier -- was slower (0.56 against 0.535 ms: that epilogue then takes 2.1 us beside the other
  // group's MFMAs; repeated in round 3 with the packed-f32 operations taken out of the epilogue, -fno-slp-vectorize: 0.523 against
  // 0.494 ms); s_setprio for either role changed nothing; both channel blocks through the patch behind ONE wait (block 1 written as
  // soon as block 0's reads are issued, eight stores in a row) was slower too: 0.549 against 0.515 ms.
#if W4_SKEW
  // Experiment: the second product's stores one interval late, so that on every SIMD this wave's vector work (DPP transposition, bias,
  // mask, stores: no LDS) runs beside the first product's MFMAs and its MFMAs beside the first product's epilogue -- matrix beside
  // vector instead of matrix beside matrix (MI355X_MICROARCH.md, "Two waves per SIMD").  The accumulators live across one barrier.
  {
    const float4 bw[2] = {ws_bias_wide(J.b2, 32 * pw, lane), ws_bias_wide(J.b2, 32 * pw + 16, lane)};
    floatx4 acc[2][4];
    auto out = [&](int k) {                                  // tile k of this workgroup leaves
      const int t = who.rank + k * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
      float* Yt = J.Y + (size_t)b * J.y_bs + p0;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) ws_store_wide(acc[cb], Yt, 32 * pw + 16 * cb, P, &bw[cb], Ms + (k & 3) * 64, P - p0, lane, false, false);
    };
    for (int i = 0; i <= nloc; ++i) {
      if (i >= 2) out(i - 2);
      if (i >= 1) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[cb][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        w4_mma8<NS, false>(acc, wh, wl, Hs + ((i - 1) & 1) * BUF, ad);
      }
      __syncthreads();
    }
    out(nloc - 1);
    return;
  }
#endif
  float b2v[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) b2v[cb] = J.b2 ? J.b2[32 * pw + 16 * cb + ci] : 0.f;
  float* Rw = Rs + pw * 16 * WS_WIDE_PITCH;
  for (int i = 0; i <= nloc; ++i) {
    if (pw == 0) W4_TRACE(2, i, 0);
    if (i >= 1) {                                            // finish tile i-1
      const int t = who.rank + (i - 1) * who.per_job;
      const int b = __builtin_amdgcn_readfirstlane(t / tpp), p0 = __builtin_amdgcn_readfirstlane((t - b * tpp) * WS_TP);
      floatx4 acc[2][4];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[cb][j] = W4_BIASINIT ? floatx4{b2v[cb], b2v[cb], b2v[cb], b2v[cb]} : floatx4{0.f, 0.f, 0.f, 0.f};
      if (!(W4_ABLATE & 2))
      w4_mma8<NS, false>(acc, wh, wl, Hs + ((i - 1) & 1) * BUF, ad);
      if (pw == 0) W4_TRACE(2, i, 2);
      const float* Mb = Ms + ((i - 1) & 3) * 64;
      float* Yt = J.Y + (size_t)b * J.y_bs + p0;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) ws_store_wide_lds<!W4_BIASINIT>(acc[cb], Rw, Yt, 32 * pw + 16 * cb, P, b2v[cb], Mb, (W4_ABLATE & 1) ? 0 : P - p0, lane);
    }
    if (pw == 0) W4_TRACE(2, i, 1);
    __syncthreads();
  }
}

template <int NS, int STG>
static int w4_mlp_launch(const WsArgs& a, hipStream_t st) {
  const size_t lds = (size_t)4 * (NS > 1 ? 2 : 1) * W3_IMG + (4 * 64 + 2 * 256 + 4 * 16 * WS_WIDE_PITCH) * sizeof(float);
  const bool nrm = a.aff != nullptr, inl = a.inl_w != nullptr;   // in_lin's output is not normalised: never both
  auto kern = inl ? w4_mlp_kernel<NS, false, true, STG> : (nrm ? w4_mlp_kernel<NS, true, false, STG> : w4_mlp_kernel<NS, false, false, STG>);
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds), "w4_mlp"));
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  const long slots = w3_slots(a.njobs, ntiles);              // one workgroup per CU: 32 per XCD, jobs that share X on one XCD
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.njobs * slots)), dim3(64 * (STG + 8)), lds, st, a);
  if (prof) prof_after(st, PROF_WS_MLP2, prof_arith_of_pieces(NS), 4.0 * WS_H * WS_H * (double)a.P * a.B * a.njobs, ws_alg_bytes(a, 1, false, false));
#if W4_TRACE_ON
  if (a.dbg & 2048) {
    static int dumps = 0;
    if (dumps++ == 20) {                                     // one warm launch
      static unsigned long long h[3 * 3 * 256];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w4_trace), sizeof(h));
      FILE* f = fopen("gpurun_out/w4_trace.txt", "w");
      if (f) { for (int i = 0; i < 256; ++i) fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", i, h[i * 9], h[i * 9 + 1], h[i * 9 + 2], h[i * 9 + 3], h[i * 9 + 4], h[i * 9 + 5], h[i * 9 + 6], h[i * 9 + 7], h[i * 9 + 8]); fclose(f); }
    }
  }
#endif
  return check_launch("w4_mlp");
}

bool w4_mlp_ok(const WsArgs& a) {
  if (a.njobs < 1 || a.njobs > WS_MAX_JOBS || a.P % 4 != 0 || (long)WS_H * a.P >= (1L << 31) || (a.maskp && !ws_al16(a.maskp))) return false;
  if (a.inl_w && (a.aff || !a.inl_in || !ws_al16(a.inl_in) || a.inl_bs % 4)) return false;
  for (int i = 0; i < a.njobs; ++i) {
    const WsJob& j = a.job[i];
    if (!j.X || !j.W1 || !j.W2 || !j.Y || j.Mid || j.accumulate || !ws_al16(j.X) || !ws_al16(j.Y) || j.x_bs % 4 || j.y_bs % 4) return false;
  }
  return true;
}

int w4_mlp_run(const WsArgs& a0, int ns, hipStream_t st) {
  WsArgs a = a0;
#if W4_TRACE_ON
  a.dbg |= sw(SW_DBG_W4);
#endif
  if (!w4_mlp_ok(a)) return fail(DPPM_E_INVALID, "w4_mlp: unsupported arguments");
  // STG = 4 only: at 16 waves (STG = 8, 128 registers) the second fragment set of w4_mma8 spills inside the multiply loop
  // (0.72 ms against 0.57 ms per pair at N = 200, B = 32)
  return ns == 3 ? w4_mlp_launch<3, 4>(a, st) : w4_mlp_launch<1, 4>(a, st);
}

template <int NS>
static int w4_lin2_launch(const WsArgs& a, hipStream_t st) {
  const size_t lds = (size_t)2 * (NS > 1 ? 2 : 1) * W3_IMG + 2 * 64 * sizeof(float);
  const bool nrm = a.stat != nullptr, inl = a.inl_w != nullptr;   // the in_lin form always comes with the statistics (layer 0 of the fused-norm forward)
  auto kern = inl ? w4_lin2_kernel<NS, true, true> : (nrm ? w4_lin2_kernel<NS, true, false> : w4_lin2_kernel<NS, false, false>);
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds), "w4_lin2"));
  const long ntiles = (long)a.B * ((a.P + WS_TP - 1) / WS_TP);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);       // one workgroup per CU
  const bool prof = prof_enabled();
  if (prof) prof_before(st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_THREADS), lds, st, a);
  if (prof) prof_after(st, PROF_WS_SUM2, prof_arith_of_pieces(NS), 4.0 * WS_H * WS_H * (double)a.P * a.B, ws_alg_bytes(a, 2, false, false));
  return check_launch("w4_lin2");
}

bool w4_lin2_ok(const WsArgs& a) {
  const WsJob& j = a.job[0];
  const bool inl = a.inl_w != nullptr;
  if (inl && !(a.inl_in && ws_al16(a.inl_in) && a.inl_bs % 4 == 0 && a.stat)) return false;
  return a.njobs == 1 && !a.side && !a.act && !a.maskp && !j.accumulate && j.X && (inl || j.X2) && j.W1 && j.W2 && j.Y && ws_al16(j.X) && (inl || ws_al16(j.X2)) &&
         ws_al16(j.Y) && j.x_bs % 4 == 0 && (inl || j.x2_bs % 4 == 0) && j.y_bs % 4 == 0 && a.P % 4 == 0 && (long)WS_H * a.P < (1L << 31) &&
         (a.stat == nullptr || (a.aff && a.cnt && a.maskstat));
}

int w4_lin2_run(const WsArgs& a, int ns, hipStream_t st) {
  if (!w4_lin2_ok(a)) return fail(DPPM_E_INVALID, "w4_lin2: unsupported arguments");
  return ns == 3 ? w4_lin2_launch<3>(a, st) : w4_lin2_launch<1>(a, st);
}

}  // namespace dppm
