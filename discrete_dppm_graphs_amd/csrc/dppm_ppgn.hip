// PPGN ("Powerful") denoiser forward/backward for gfx950 behind the C-ABI of include/dppm_hip.h.
//
// Data layout in HBM: every activation is a stack of channel-first planes [B][C][N][ld] (ld = N rounded up to 4,
// P = N*ld floats per plane, pad columns hold zeros wherever a tensor is masked).  In this layout
//   - each nn.Linear of the reference (applied at every (i,j), model/ppgn.py:58-59,69,301,327-333) is a plain
//     row-major GEMM  Y[b] (O x P) = W (O x I) * X[b] (I x P)  with the weight in torch's own [out][in] layout,
//   - the per-channel contraction out1 @ out2 (model/ppgn.py:62) reads and writes contiguous N x ld planes,
//   - the masked instance norm (util/model_helper.py:176-191) reduces over one contiguous plane,
//   - torch.cat along features (model/ppgn.py:66,318) is adjacency of planes: the skip-concat Linear is a
//     two-segment K loop and the readout reads the layer outputs where they already lie (buffer CAT).
// so the reference's permutes and cats never cost a byte.  All GEMMs run on v_mfma_f32_32x32x2_f32 (dppm_gemm.h).
#include "dppm_gemm.h"
#include "dppm_wstat.h"
#include "dppm_wsplit.h"
#include "dppm_head.h"
#include "dppm_w4.h"
#include "dppm_w6.h"
#include "dppm_plane4.h"
#include "dppm_wgrad.h"
#include "dppm_planemm.h"
#include "dppm_planesplit.h"
#include "dppm_smallgraph.h"
#include <cstdlib>

namespace dppm {
namespace {

// ---------------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------------
struct Layout {
  int B, N, H, L, ld, P, CT;       // CT = 2 + L*H channels of CAT
  size_t S;                        // floats in one [B][H][P] plane set
  size_t maskp, inv_norm, cat, u0, R, act, tmp[4];
  size_t hid1, hid2, o1, o2, mm;   // per-layer saves (training): base + l * S
  size_t rstd;                     // [L][B][H]
  size_t himg;                     // inference, H = 128: bf16 weight fragments of the fused read-out head (dppm_head.h)
  size_t nstat, ncnt, naff;        // inference, H = 128 / 32: per-tile statistics [B*tpp][H][2], counts [B*tpp], (scale, shift) tables [L+1][B][2][H]
  int w32_jobs;
  size_t dh, dz, dcat, dy, dmm, do1, do2, du0, slabs, rsum, w32;   // w32: slabs + row sums of up to six 32 x 32 weight gradients (H = 32)
  size_t slab_floats, total;
  int training;
};

constexpr size_t kAlign = 64;   // floats (256 B)
constexpr int W32_DEFER_MAXL = 16, W32_DEFER_MAXNZ = 512;   // small networks: per-layer weight-gradient slabs, folded by one launch (wgrad32_multi)
inline size_t bump(size_t& off, size_t n) {
  const size_t at = off;
  off += (n + kAlign - 1) / kAlign * kAlign;
  return at;
}

inline int wgrad_nsplit(int B, int P) {   // K chunks per graph so that ~1024 workgroups exist
  int per = cdiv(1024, B);
  int kchunk = round_up(cdiv(P, per), 256);
  if (kchunk < 256) kchunk = 256;
  return cdiv(P, kchunk);
}
inline int wgrad_kchunk(int B, int P) { return round_up(cdiv(P, wgrad_nsplit(B, P)), BK); }

Layout make_layout(int B, int N, int H, int L, int training) {
  Layout w{};
  w.B = B; w.N = N; w.H = H; w.L = L; w.training = training;
  w.ld = round_up(N, 4);
  w.P = N * w.ld;
  w.CT = 2 + L * H;
  w.S = (size_t)B * H * w.P;
  size_t off = 0;
  w.maskp = bump(off, (size_t)B * w.P);
  w.inv_norm = bump(off, B);
  w.cat = bump(off, (size_t)B * w.CT * w.P);
  w.u0 = bump(off, w.S);
  w.R = bump(off, w.S);
  w.act = bump(off, w.S);
  w.rstd = bump(off, (size_t)L * B * H);
  if (!training) {
    for (int i = 0; i < 4; ++i) w.tmp[i] = bump(off, w.S);
    w.himg = H == WS_H ? bump(off, head3_wimg_floats(L)) : 0;
    if (H == WS_H || H == 32) {   // instance norm folded into its producer and consumers (H = 32: the wide kernels)
      const size_t ntiles = (size_t)B * ((w.P + WS_TP - 1) / WS_TP);
      w.nstat = bump(off, ntiles * H * 2);
      w.ncnt = bump(off, ntiles);
      w.naff = bump(off, (size_t)(L + 1) * B * 2 * H);
    }
  } else {
    w.tmp[3] = bump(off, w.S);   // y
    w.hid1 = bump(off, w.S * L); w.hid2 = bump(off, w.S * L);
    w.o1 = bump(off, w.S * L); w.o2 = bump(off, w.S * L); w.mm = bump(off, w.S * L);
    w.dh = bump(off, w.S); w.dz = bump(off, w.S);
    w.dcat = bump(off, (size_t)B * w.CT * w.P);
    w.dy = bump(off, w.S); w.dmm = bump(off, w.S); w.do1 = bump(off, w.S); w.do2 = bump(off, w.S);
    w.du0 = bump(off, w.S);
    const int nz = B * wgrad_nsplit(B, w.P);
    w.slab_floats = (size_t)nz * H * (size_t)(w.CT > 2 * H ? w.CT : 2 * H);
    if (H == WS_H && w.slab_floats < ws_wgrad_scratch_floats()) w.slab_floats = ws_wgrad_scratch_floats();
    w.slabs = bump(off, w.slab_floats);
    w.rsum = bump(off, (size_t)nz * H + (size_t)B * H + B);
    // few slabs (small graphs) and few layers: every weight-gradient launch keeps its own slabs, ONE reduction launch folds them all
    // (8 = W32_MAXJOBS; deferred: six jobs per layer, the head's L + 2 and in_lin's one)
    w.w32_jobs = (H == 32 && nz <= W32_DEFER_MAXNZ && L <= W32_DEFER_MAXL) ? 7 * L + 3 : 8;
    w.w32 = H == 32 ? bump(off, (size_t)w.w32_jobs * nz * (1024 + 32)) : 0;
  }
  w.total = off;
  return w;
}

// ---------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------

// Input planes: CAT[b][0] = A (padded), CAT[b][1] = sigma_b on the diagonal (model/ppgn.py:283-293), the padded mask
// and 1/sqrt(#active nodes) from mask row 0 (model/ppgn.py:51).
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ A, const float* __restrict__ mask,
                                                   const float* __restrict__ sigma, int N, int ld, int CT,
                                                   float* __restrict__ cat, float* __restrict__ maskp,
                                                   float* __restrict__ inv_norm) {
  const int b = blockIdx.y;
  const int P = N * ld;
  float* c0 = cat + (size_t)b * CT * P;
  float* c1 = c0 + P;
  const float sg = sigma[b];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int i = p / ld, j = p % ld;
    const bool in = j < N;
    const size_t src = ((size_t)b * N + i) * N + j;
    c0[p] = in ? A[src] : 0.f;
    c1[p] = (in && i == j) ? sg : 0.f;
    maskp[(size_t)b * P + p] = in ? mask[src] : 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 64) s += mask[(size_t)b * N * N + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) inv_norm[b] = 1.f / sqrtf(s);
  }
}

__device__ __forceinline__ float block_sum_1024(float v, float* sh) {   // all threads get the total
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = blockDim.x >> 6;
  for (int k = 0; k < nw; ++k) t += sh[k];
  return t;
}

constexpr int NORM_THREADS = 1024;   // large planes; small ones use 256 or 64 threads (norm_threads below)
constexpr int NORM_CACHE = 10;   // float4 per thread kept in registers: planes up to 40960 floats stay on chip

// masked instance norm, one workgroup per (b,c) plane: mean, then variance around it, then normalise
// (two-pass statistics as in util/model_helper.py:182-189; the plane is read from HBM once and kept in registers)
template <int NTH>
__global__ __launch_bounds__(NTH) void inorm_fwd_kernel(const float* __restrict__ y, long y_bs,
                                                                 const float* __restrict__ maskp, float* __restrict__ u,
                                                                 long u_bs, int C, int P, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out) {
  __shared__ float sh[NTH / 64];
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  const float4* yp = reinterpret_cast<const float4*>(y + (size_t)b * y_bs + (size_t)c * P);
  const float4* mp = reinterpret_cast<const float4*>(maskp + (size_t)b * P);
  float4* up = reinterpret_cast<float4*>(u + (size_t)b * u_bs + (size_t)c * P);
  const int P4 = P / 4;
  float4 cy[NORM_CACHE], cm[NORM_CACHE];
  float s = 0.f, cnt = 0.f;
#pragma unroll
  for (int v = 0; v < NORM_CACHE; ++v) {
    const int idx = threadIdx.x + v * NTH;
    cy[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    cm[v] = cy[v];
    if (idx < P4) { cy[v] = yp[idx]; cm[v] = mp[idx]; }
    s += cy[v].x * cm[v].x + cy[v].y * cm[v].y + cy[v].z * cm[v].z + cy[v].w * cm[v].w;
    cnt += cm[v].x + cm[v].y + cm[v].z + cm[v].w;
  }
  for (int idx = threadIdx.x + NORM_CACHE * NTH; idx < P4; idx += NTH) {
    const float4 a = yp[idx], m = mp[idx];
    s += a.x * m.x + a.y * m.y + a.z * m.z + a.w * m.w;
    cnt += m.x + m.y + m.z + m.w;
  }
  s = block_sum_1024(s, sh);
  cnt = block_sum_1024(cnt, sh);
  const float mean = s / cnt;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NORM_CACHE; ++v) {
    const float dx = (cy[v].x - mean) * cm[v].x, dy = (cy[v].y - mean) * cm[v].y;
    const float dz = (cy[v].z - mean) * cm[v].z, dw = (cy[v].w - mean) * cm[v].w;
    q += dx * dx + dy * dy + dz * dz + dw * dw;
  }
  for (int idx = threadIdx.x + NORM_CACHE * NTH; idx < P4; idx += NTH) {
    const float4 a = yp[idx], m = mp[idx];
    const float dx = (a.x - mean) * m.x, dy = (a.y - mean) * m.y, dz = (a.z - mean) * m.z, dw = (a.w - mean) * m.w;
    q += dx * dx + dy * dy + dz * dz + dw * dw;
  }
  q = block_sum_1024(q, sh);
  const float rstd = 1.f / sqrtf(q / cnt + 1e-5f);
#pragma unroll
  for (int v = 0; v < NORM_CACHE; ++v) {
    const int idx = threadIdx.x + v * NTH;
    if (idx < P4)
      up[idx] = make_float4((cy[v].x - mean) * rstd * cm[v].x, (cy[v].y - mean) * rstd * cm[v].y,
                            (cy[v].z - mean) * rstd * cm[v].z, (cy[v].w - mean) * rstd * cm[v].w);
  }
  for (int idx = threadIdx.x + NORM_CACHE * NTH; idx < P4; idx += NTH) {
    const float4 a = yp[idx], m = mp[idx];
    up[idx] = make_float4((a.x - mean) * rstd * m.x, (a.y - mean) * rstd * m.y, (a.z - mean) * rstd * m.z,
                          (a.w - mean) * rstd * m.w);
  }
  if (threadIdx.x == 0) {
    if (mean_out) mean_out[blockIdx.x] = mean;
    if (rstd_out) rstd_out[blockIdx.x] = rstd;
  }
}

// dy = rstd * m * (g - mean_m(g) - (u - mean_m(u)) * mean_m(g*u))   (u already carries the mask).
// mean_m(u) is zero in exact arithmetic; keeping it makes sum(dy) vanish structurally, as it does in the autograd
// graph of util/model_helper.py:182-190 (the rounding of the forward mean otherwise leaks a constant into dy).
template <int NTH>
__global__ __launch_bounds__(NTH) void inorm_bwd_kernel(const float* __restrict__ g, long g_bs,
                                                                 const float* __restrict__ u, long u_bs,
                                                                 const float* __restrict__ maskp,
                                                                 const float* __restrict__ rstd_in, float* __restrict__ dy,
                                                                 int C, int P) {
  __shared__ float sh[NTH / 64];
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  const float4* gp = reinterpret_cast<const float4*>(g + (size_t)b * g_bs + (size_t)c * P);
  const float4* up = reinterpret_cast<const float4*>(u + (size_t)b * u_bs + (size_t)c * P);
  const float4* mp = reinterpret_cast<const float4*>(maskp + (size_t)b * P);
  float4* dp = reinterpret_cast<float4*>(dy + ((size_t)b * C + c) * P);
  const int P4 = P / 4;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f, cnt = 0.f;
  for (int idx = threadIdx.x; idx < P4; idx += NTH) {
    const float4 a = gp[idx], x = up[idx], m = mp[idx];
    s1 += a.x * m.x + a.y * m.y + a.z * m.z + a.w * m.w;
    s2 += a.x * x.x + a.y * x.y + a.z * x.z + a.w * x.w;
    s3 += (x.x + x.y) + (x.z + x.w);
    cnt += m.x + m.y + m.z + m.w;
  }
  s1 = block_sum_1024(s1, sh);
  s2 = block_sum_1024(s2, sh);
  s3 = block_sum_1024(s3, sh);
  cnt = block_sum_1024(cnt, sh);
  const float rstd = rstd_in[blockIdx.x];
  const float m1 = s1 / cnt, m2 = s2 / cnt, ubar = s3 / cnt;
  for (int idx = threadIdx.x; idx < P4; idx += NTH) {
    const float4 a = gp[idx], x = up[idx], m = mp[idx];
    dp[idx] = make_float4(rstd * m.x * (a.x - m1 - (x.x - ubar) * m2), rstd * m.y * (a.y - m1 - (x.y - ubar) * m2),
                          rstd * m.z * (a.z - m1 - (x.z - ubar) * m2), rstd * m.w * (a.w - m1 - (x.w - ubar) * m2));
  }
}

// workgroup size for planes of P floats: one float4 per thread and pass up to NORM_CACHE passes stay in registers; a 1024-thread
// group on a 360-float plane (N = 18) spends its time in the three block reductions (22 us per launch, measured)
inline int norm_threads(int P) { return P <= 512 ? 64 : (P <= 4096 ? 256 : NORM_THREADS); }

template <typename... Args>
inline void launch_inorm_fwd(int P, int nplanes, hipStream_t st, Args... args) {
  const int nth = norm_threads(P);
  if (nth == 64) hipLaunchKernelGGL(inorm_fwd_kernel<64>, dim3(nplanes), dim3(64), 0, st, args...);
  else if (nth == 256) hipLaunchKernelGGL(inorm_fwd_kernel<256>, dim3(nplanes), dim3(256), 0, st, args...);
  else hipLaunchKernelGGL(inorm_fwd_kernel<NORM_THREADS>, dim3(nplanes), dim3(NORM_THREADS), 0, st, args...);
}
template <typename... Args>
inline void launch_inorm_bwd(int P, int nplanes, hipStream_t st, Args... args) {
  const int nth = norm_threads(P);
  if (nth == 64) hipLaunchKernelGGL(inorm_bwd_kernel<64>, dim3(nplanes), dim3(64), 0, st, args...);
  else if (nth == 256) hipLaunchKernelGGL(inorm_bwd_kernel<256>, dim3(nplanes), dim3(256), 0, st, args...);
  else hipLaunchKernelGGL(inorm_bwd_kernel<NORM_THREADS>, dim3(nplanes), dim3(NORM_THREADS), 0, st, args...);
}

// Instance norm folded into its producer and consumers (split-bf16 inference, dppm_wsplit.h): the two-source Linear m4 leaves, per
// 64-position tile and channel, (masked mean, centred sum of squares) and the tile's unmasked count; this kernel merges the tiles
// of a graph into (scale, shift) = (rstd, -mean * rstd) with the reference's biased variance and eps = 1e-5
// (util/model_helper.py:182-190):  mean = sum n_t mean_t / n,  M2 = sum (M2_t + n_t (mean_t - mean)^2), accumulated in double.
// One workgroup per (graph, 16 channels): 64 tile groups x 16 channels (a tile's 16 float2 are 128 contiguous bytes).
constexpr int NF_CH = 16, NF_GRP = 64;
__global__ __launch_bounds__(NF_CH * NF_GRP) void norm_finalize_kernel(const float* __restrict__ stat, const float* __restrict__ cnt, int tpp,
                                                                       float* __restrict__ aff, int C) {   // C channels (a multiple of NF_CH); grid B * C / NF_CH
  __shared__ double s0[NF_GRP][NF_CH], s1[NF_GRP][NF_CH];
  __shared__ double tot_n[NF_CH], tot_m[NF_CH];
  const int b = blockIdx.x / (C / NF_CH), c0 = (blockIdx.x % (C / NF_CH)) * NF_CH;
  const int c = threadIdx.x % NF_CH, grp = threadIdx.x / NF_CH;
  const size_t base = (size_t)b * tpp;
  // pass 1: n and n * mean
  double n = 0.0, sm = 0.0;
  for (int t = grp; t < tpp; t += NF_GRP) {
    const double nt = cnt[base + t];
    const float2 s = *reinterpret_cast<const float2*>(stat + ((base + t) * C + c0 + c) * 2);
    n += nt;
    sm += nt > 0.0 ? nt * (double)s.x : 0.0;
  }
  s0[grp][c] = n; s1[grp][c] = sm;
  __syncthreads();
  if (grp == 0) {
    double a0 = 0.0, a1 = 0.0;
    for (int k = 0; k < NF_GRP; ++k) { a0 += s0[k][c]; a1 += s1[k][c]; }
    tot_n[c] = a0; tot_m[c] = a0 > 0.0 ? a1 / a0 : 0.0;
  }
  __syncthreads();
  const double mean = tot_m[c];
  // pass 2: centred sum of squares about the graph's mean
  double m2 = 0.0;
  for (int t = grp; t < tpp; t += NF_GRP) {
    const double nt = cnt[base + t];
    const float2 s = *reinterpret_cast<const float2*>(stat + ((base + t) * C + c0 + c) * 2);
    const double d = (double)s.x - mean;
    m2 += nt > 0.0 ? (double)s.y + nt * d * d : 0.0;
  }
  s0[grp][c] = m2;
  __syncthreads();
  if (grp == 0) {
    double a0 = 0.0;
    for (int k = 0; k < NF_GRP; ++k) a0 += s0[k][c];
    const double nn = tot_n[c];
    const float var = nn > 0.0 ? (float)(a0 / nn) : 0.f;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    aff[(size_t)b * 2 * C + c0 + c] = rstd;
    aff[(size_t)b * 2 * C + C + c0 + c] = (float)(-mean) * rstd;
  }
}
// debug (DPPM_DBG & 4): the table from the standalone pass's own statistics
__global__ __launch_bounds__(256) void aff_from_stats_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ aff, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (b, c)
  if (i < n) { const int b = i / WS_H, c = i % WS_H; aff[(size_t)b * 2 * WS_H + c] = rstd[i]; aff[(size_t)b * 2 * WS_H + WS_H + c] = -mean[i] * rstd[i]; }
}
// the identity table (layer 0's skip input u0 is not normalised)
__global__ __launch_bounds__(256) void aff_identity_kernel(float* __restrict__ aff, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) aff[i] = (i & 255) < WS_H ? 1.f : 0.f;
}

// Powerful_Ind conditioning: plane[b][i][i] = cond[b] for i < N on the (all-zero) pad channel of a layer's input
__global__ __launch_bounds__(256) void diag_set_kernel(float* __restrict__ plane, long bs, int N, int ld, const float* __restrict__ cond) {
  const int b = blockIdx.x;
  const float v = cond[b];
  for (int i = threadIdx.x; i < N; i += blockDim.x) plane[(size_t)b * bs + (size_t)i * ld + i] = v;
}

// d layer_cond[b] = sum_i dX[b][pad channel][i][i]: the conditioning scalar sits on that channel's diagonal (diag_set_kernel)
__global__ __launch_bounds__(64) void diag_sum_kernel(const float* __restrict__ plane, long bs, int N, int ld, float* __restrict__ out) {
  const int b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 64) s += plane[(size_t)b * bs + (size_t)i * ld + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[b] = s;
}

// d sigma[b]: sigma sits on the diagonal of input channel 1 (prep_kernel), which feeds in_lin (gradient du0) and the readout
// layer_cat_lin (gradient dR):  sum_i sum_c ( in_w[c][1] du0[b][c][ii] + cat_w[c][1] dR[b][c][ii] )
__global__ __launch_bounds__(256) void dsigma_kernel(const float* __restrict__ du0, const float* __restrict__ dR, const float* __restrict__ in_w,
                                                     const float* __restrict__ cat_w, int CT, int H, int N, int ld, float* __restrict__ out) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const size_t P = (size_t)N * ld;
  float s = 0.f;
  for (int e = threadIdx.x; e < H * N; e += 256) {
    const int c = e / N, i = e % N;
    const size_t at = ((size_t)b * H + c) * P + (size_t)i * ld + i;
    s += in_w[2 * c + 1] * du0[at] + cat_w[(size_t)c * CT + 1] * dR[at];
  }
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) out[b] = s;
}

// logits[b][i][j] = sum_o wfin[o] * (R + act)[b][o][p] + bfin     (model/ppgn.py:331-333; dropout is identity in eval)
__global__ __launch_bounds__(256) void final_fwd_kernel(const float* __restrict__ R, const float* __restrict__ act,
                                                        const float* __restrict__ wfin, const float* __restrict__ bfin,
                                                        int H, int N, int ld, float* __restrict__ logits) {
  const int b = blockIdx.y;
  const int P = N * ld;
  const float bias = bfin[0];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int i = p / ld, j = p % ld;
    if (j >= N) continue;
    float acc = 0.f;
    const float* r = R + (size_t)b * H * P + p;
    const float* a = act + (size_t)b * H * P + p;
    for (int o = 0; o < H; ++o) acc += wfin[o] * (r[(size_t)o * P] + a[(size_t)o * P]);
    logits[((size_t)b * N + i) * N + j] = acc + bias;
  }
}

// Backward of the head, one workgroup per (b,o) plane:
//   dh = wfin[o] * dl ; dz = dh * LeakyReLU'(act) ; dwfin partial = sum_p (R+act) * dl ; dbfin partial = sum_p dl
__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ R, const float* __restrict__ act,
                                                        const float* __restrict__ wfin, const float* __restrict__ dlogits,
                                                        int H, int N, int ld, float* __restrict__ dh, float* __restrict__ dz,
                                                        float* __restrict__ dw_part, float* __restrict__ db_part) {
  __shared__ float sh[4];
  const int b = blockIdx.x / H, o = blockIdx.x % H;
  const int P = N * ld;
  const size_t base = ((size_t)b * H + o) * P;
  const float w = wfin[o];
  float sw = 0.f, sb = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int i = p / ld, j = p % ld;
    const float dl = j < N ? dlogits[((size_t)b * N + i) * N + j] : 0.f;
    const float a = act[base + p];
    const float h = R[base + p] + a;
    const float d = w * dl;
    dh[base + p] = d;
    dz[base + p] = d * (a > 0.f ? 1.f : kSlope);
    sw += h * dl;
    sb += dl;
  }
  sw = block_sum_1024(sw, sh);
  sb = block_sum_1024(sb, sh);
  if (threadIdx.x == 0) {
    dw_part[blockIdx.x] = sw;
    if (o == 0) db_part[b] = sb;
  }
}

// out[m*ldo + n] = sum_z slabs[z][m*N + n]   (fixed order: bitwise reproducible).  A block folds 32 consecutive elements: eight
// groups of 32 threads take every eighth slab (coalesced 128-byte rows), then the eight partial sums are added in order.  The
// former one-thread-per-element loop over all slabs was a chain of nz dependent loads on a handful of blocks: 15 us per call
// at 32 x 32 outputs, 74 calls per training step of the small (6 x 32) networks = 40 % of that step.
constexpr int RS_E = 32, RS_G = 8;
// second (nullable) job in the same launch: out2[m] = sum_z slabs2[z][m] (the bias gradient that rides with a weight gradient)
__global__ __launch_bounds__(RS_E * RS_G) void reduce_slabs_kernel(const float* __restrict__ slabs, int nz, int M, int N,
                                                                 float* __restrict__ out, int ldo,
                                                                 const float* __restrict__ slabs2, float* __restrict__ out2) {
  __shared__ float part[RS_G][RS_E];
  const int g = threadIdx.x / RS_E, t = threadIdx.x % RS_E;
  int e = blockIdx.x * RS_E + t;
  int MN = M * N;
  const int nb1 = (MN + RS_E - 1) / RS_E;            // blocks of the first job
  const bool second = (int)blockIdx.x >= nb1;        // block-uniform
  if (second) { e -= nb1 * RS_E; slabs = slabs2; MN = M; }
  float acc = 0.f;
  if (e < MN)
    for (int z = g; z < nz; z += RS_G) acc += slabs[(size_t)z * MN + e];
  part[g][t] = acc;
  __syncthreads();
  if (g == 0 && e < MN) {
    float v = part[0][t];
#pragma unroll
    for (int k = 1; k < RS_G; ++k) v += part[k][t];
    if (second) out2[e] = v;
    else out[(size_t)(e / N) * ldo + (e % N)] = v;
  }
}

// Weight gradient of a 32 x 32 Linear (the 6 x 32 networks of ego_small / community_small), one K chunk of one graph per workgroup:
//   slab[z] (32 x 32) = dY[b][:, k0 .. k0+kc) * X[b][:, k0 .. k0+kc)^T,  rsum[z][m] = sum_k dY[b][m][k]      (z = b * nsplit + s)
// Both operand chunks (2 x 32 x kc floats, kc <= 512) are brought into LDS with every load in flight at once, then four waves
// take one 16 x 16 output tile each on v_mfma_f32_16x16x4_f32.  The generic split-K GEMM walks the same chunk in 16-position steps
// of load -> LDS -> barrier -> MFMA on a 32 x 128 tile: 16.6 us per launch for this shape, 40 % of a small-network training step.
constexpr int W32_MAXK = 512;
__device__ __forceinline__ void wgrad32_body(const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs,
                                             int P, int nsplit, int kchunk, float* __restrict__ slabs, float* __restrict__ rsum, int z) {
  extern __shared__ __attribute__((aligned(16))) float w32_lds[];
  const int b = z / nsplit, s = z % nsplit;
  const int k0 = s * kchunk, kc = min(kchunk, P - k0);          // kchunk % 16 == 0, P % 4 == 0
  const int pitch = kchunk + 4;                                   // == 4 mod 16: fragment reads of 16 rows x 4 k are conflict-free
  float* sA = w32_lds;
  float* sB = w32_lds + 32 * pitch;
  const float* gA = dY + (size_t)b * dy_bs + k0;
  const float* gB = X + (size_t)b * x_bs + k0;
  const int kq = kchunk / 4;
  for (int e0 = threadIdx.x; e0 < 32 * kq; e0 += 4 * 256) {     // four float4 per operand in flight per thread before any LDS store
    float4 va[4], vb[4];
    int off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 256 * u;
      const int r = e / kq, c4 = (e - r * kq) * 4;
      off[u] = e < 32 * kq ? r * pitch + c4 : -1;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (off[u] >= 0 && c4 < kc) {
        va[u] = *reinterpret_cast<const float4*>(gA + (size_t)r * P + c4);
        vb[u] = *reinterpret_cast<const float4*>(gB + (size_t)r * P + c4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (off[u] >= 0) {
        *reinterpret_cast<float4*>(sA + off[u]) = va[u];
        *reinterpret_cast<float4*>(sB + off[u]) = vb[u];
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int m0 = 16 * (w >> 1), n0 = 16 * (w & 1);
  const float* pa = sA + (m0 + n) * pitch + q;
  const float* pb = sB + (n0 + n) * pitch + q;
  floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f}, rs = acc;
  const bool do_rs = rsum != nullptr && n0 == 0;                  // wave-uniform
  for (int k = 0; k < kchunk; k += 4) {
    const float av = pa[k];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, pb[k], acc, 0, 0, 0);
    if (do_rs) rs = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.f, rs, 0, 0, 0);   // every column = the row sums
  }
  float* out = slabs + (size_t)z * 1024;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(m0 + 4 * q + r) * 32 + n0 + n] = acc[r];
  if (do_rs && n == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) rsum[(size_t)z * 32 + m0 + 4 * q + r] = rs[r];
  }
}
__global__ __launch_bounds__(256) void wgrad32_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs,
                                                      int P, int nsplit, int kchunk, float* __restrict__ slabs,
                                                      float* __restrict__ rsum) {
  wgrad32_body(dY, dy_bs, X, x_bs, P, nsplit, kchunk, slabs, rsum, blockIdx.x);
}

// Up to six independent 32 x 32 weight gradients in one launch (ALL of a layer's: m4's halves, both MLPs' matrices), and their slab
// reductions in one more: the per-matrix pairs of launches were 47 % of a small-network training step.
constexpr int W32_MAXJOBS = 8;
struct W32Job { const float* dY; long dy_bs; const float* X; long x_bs; float* dW; int ldo; float* db; int ncols; };   // ncols: columns of the 32 x 32 block that are stored (0 = all 32)
struct W32Args { W32Job job[W32_MAXJOBS]; int njobs, P, nsplit, kchunk, nz; float* slabs; float* rsum; };   // slabs [job][nz][1024], rsum [job][nz][32]
__global__ __launch_bounds__(256) void wgrad32m_kernel(const W32Args a) {
  const int j = blockIdx.x / a.nz, z = blockIdx.x - j * a.nz;
  const W32Job& q = a.job[j];
  wgrad32_body(q.dY, q.dy_bs, q.X, q.x_bs, a.P, a.nsplit, a.kchunk, a.slabs + (size_t)j * a.nz * 1024, a.rsum + (size_t)j * a.nz * 32, z);
}
// The fold of the slabs, for the jobs of one launch above or (small graphs: W32_DEFER_*) for those of ALL layers of a backward pass at once:
// job j's slabs at slabs + j * nz * 1024, its row sums at rsum + j * nz * 32
constexpr int R32_MAXJOBS = 7 * W32_DEFER_MAXL + 3;
struct R32Out { float* dW; int ldo; float* db; int ncols; };
struct R32Args { R32Out job[R32_MAXJOBS]; int njobs, nz; const float* slabs; const float* rsum; };
// per job: 32 blocks fold dW (32 outputs each), one block folds db; fixed order as reduce_slabs_kernel
__global__ __launch_bounds__(RS_E * RS_G) void reduce32m_kernel(const R32Args a) {
  __shared__ float part[RS_G][RS_E];
  const int j = blockIdx.x / 33, kind = blockIdx.x - j * 33;
  const R32Out q = a.job[j];
  const bool bias = kind == 32;                                    // block-uniform
  const int g = threadIdx.x / RS_E, t = threadIdx.x % RS_E;
  const float* src = bias ? a.rsum + (size_t)j * a.nz * 32 : a.slabs + (size_t)j * a.nz * 1024;
  const int MN = bias ? 32 : 1024, e = bias ? t : kind * RS_E + t;
  float acc = 0.f;
  for (int z = g; z < a.nz; z += RS_G) acc += src[(size_t)z * MN + e];
  part[g][t] = acc;
  __syncthreads();
  if (g == 0) {
    float v = part[0][t];
#pragma unroll
    for (int k = 1; k < RS_G; ++k) v += part[k][t];
    if (bias) { if (q.db) q.db[e] = v; }
    else if ((e & 31) < (q.ncols ? q.ncols : 32)) q.dW[(size_t)(e >> 5) * q.ldo + (e & 31)] = v;
  }
}

// Data gradient of a 32 x 32 Linear of the small networks: Y[b][o][p] (+)= (sum_c W[c][o] * X[b][c][p]) * act'(dact[b][o][p]).
// One wave per 16-position tile; the tile arrives in the MFMA result layout (lane (q, n): channels 16 ci + 4 q + r of position n,
// the trick of dppm_smallgraph.hip), every load of the wave is in flight at once, 16 MFMAs, 64-byte coalesced stores.  The generic
// GEMM spends 11 us per launch on this shape (32 x 128 tile, K loop through LDS); 37 launches per training step.
__global__ __launch_bounds__(256) void lin32t_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ X, long x_bs,
                                                     const float* __restrict__ dact, long dact_bs, float* __restrict__ Y, long y_bs,
                                                     int P, int B, int accumulate,
                                                     const float* __restrict__ W2, float* __restrict__ Y2, long y2_bs, int accumulate2) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int ntile = (P + 15) >> 4;
  const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (id >= B * ntile) return;
  const int b = id / ntile, pos = 16 * (id - b * ntile) + n;
  const bool ok = pos < P;
  const float* xb = X + (size_t)b * x_bs;
  float* yb = Y + (size_t)b * y_bs;
  floatx4 src[2], acc[2], da[2], old[2];
  float wv[2][4][2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * ci + 4 * q + r;
      src[ci][r] = ok ? xb[(size_t)c * P + pos] : 0.f;
      da[ci][r] = (dact && ok) ? dact[(size_t)b * dact_bs + (size_t)c * P + pos] : 1.f;
      old[ci][r] = (accumulate && ok) ? yb[(size_t)c * P + pos] : 0.f;
#pragma unroll
      for (int co = 0; co < 2; ++co) wv[ci][r][co] = W[(size_t)c * ldw + 16 * co + n];
    }
  acc[0] = acc[1] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int co = 0; co < 2; ++co) acc[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ci][r][co], src[ci][r], acc[co], 0, 0, 0);
  if (ok) {
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[co][r];
        if (dact) v *= da[co][r] > 0.f ? 1.f : kSlope;
        yb[(size_t)(16 * co + 4 * q + r) * P + pos] = v + old[co][r];
      }
  }
  if (W2) {   // second job on the same source (the two halves of m4^T: d mm and d x from one read of dy); same pitch ldw
    float* y2 = Y2 + (size_t)b * y2_bs;
    floatx4 acc2[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}}, old2[2];
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) old2[co][r] = (accumulate2 && ok) ? y2[(size_t)(16 * co + 4 * q + r) * P + pos] : 0.f;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co)
          acc2[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(W2[(size_t)(16 * ci + 4 * q + r) * ldw + 16 * co + n], src[ci][r], acc2[co], 0, 0, 0);
    if (ok) {
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) y2[(size_t)(16 * co + 4 * q + r) * P + pos] = acc2[co][r] + old2[co][r];
    }
  }
}

// The same transposed product for nblk consecutive 32-column blocks of one matrix against ONE source (layer_cat_lin^T dR -> the
// gradient of every layer output, small networks): Y[b][32 j + o][p] = sum_c W[c][32 j + o] * X[b][c][p].  The source tile is read
// once; the generic GEMM took 16.6 us for this 194 x 32 product.
__global__ __launch_bounds__(256) void lin32t_blocks_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ X, long x_bs,
                                                            float* __restrict__ Y, long y_bs, int P, int B, int nblk) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int ntile = (P + 15) >> 4;
  const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (id >= B * ntile) return;
  const int b = id / ntile, pos = 16 * (id - b * ntile) + n;
  const bool ok = pos < P;
  const float* xb = X + (size_t)b * x_bs;
  floatx4 src[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) src[ci][r] = ok ? xb[(size_t)(16 * ci + 4 * q + r) * P + pos] : 0.f;
  for (int j = 0; j < nblk; ++j) {
    const float* Wj = W + 32 * j;
    float* yb = Y + (size_t)b * y_bs + (size_t)32 * j * P;
    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co)
          acc[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wj[(size_t)(16 * ci + 4 * q + r) * ldw + 16 * co + n], src[ci][r], acc[co], 0, 0, 0);
    if (ok) {
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) yb[(size_t)(16 * co + 4 * q + r) * P + pos] = acc[co][r];
    }
  }
}

// Backward of both per-position MLPs of a small-network layer in one pass (32 x 32 matrices, torch [out][in] layout, pitch 32):
//   dpre_k = (W1_k^T dout_k) * LeakyReLU'(hid_k)   (stored: the first-layer weight gradients read it),   dx += sum_k W0_k^T dpre_k.
// One wave per 16-position tile; dpre_k stays in the MFMA result layout and feeds the second product from registers
// (dppm_smallgraph.hip's chaining), so the four launches of the layer-by-layer form (11 us each on the generic GEMM) become one.
__global__ __launch_bounds__(256) void mlp32_bwd_kernel(const float* __restrict__ do1, const float* __restrict__ do2,
                                                        const float* __restrict__ hid1, const float* __restrict__ hid2, long hs,
                                                        const float* __restrict__ W1a, const float* __restrict__ W1b,
                                                        const float* __restrict__ W0a, const float* __restrict__ W0b,
                                                        float* __restrict__ dpre1, float* __restrict__ dpre2,
                                                        float* __restrict__ dx, long dx_bs, int P, int B) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int ntile = (P + 15) >> 4;
  const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (id >= B * ntile) return;
  const int b = id / ntile, pos = 16 * (id - b * ntile) + n;
  const bool ok = pos < P;
  const size_t g = (size_t)b * hs;
  floatx4 accx[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}}, old[2];
#pragma unroll
  for (int co = 0; co < 2; ++co)
#pragma unroll
    for (int r = 0; r < 4; ++r) old[co][r] = ok ? dx[(size_t)b * dx_bs + (size_t)(16 * co + 4 * q + r) * P + pos] : 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* dout = k ? do2 : do1;
    const float* hid = k ? hid2 : hid1;
    const float* W1 = k ? W1b : W1a;
    const float* W0 = k ? W0b : W0a;
    float* dpre = k ? dpre2 : dpre1;
    floatx4 src[2], hv[2], t[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    float w1[2][4][2], w0[2][4][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * ci + 4 * q + r;
        src[ci][r] = ok ? dout[g + (size_t)c * P + pos] : 0.f;
        hv[ci][r] = ok ? hid[g + (size_t)c * P + pos] : 1.f;
#pragma unroll
        for (int co = 0; co < 2; ++co) { w1[ci][r][co] = W1[c * 32 + 16 * co + n]; w0[ci][r][co] = W0[c * 32 + 16 * co + n]; }
      }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co) t[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[ci][r][co], src[ci][r], t[co], 0, 0, 0);
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t[co][r] *= hv[co][r] > 0.f ? 1.f : kSlope;
        if (ok) dpre[g + (size_t)(16 * co + 4 * q + r) * P + pos] = t[co][r];
      }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co) accx[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[ci][r][co], t[ci][r], accx[co], 0, 0, 0);
  }
  if (ok) {
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) dx[(size_t)b * dx_bs + (size_t)(16 * co + 4 * q + r) * P + pos] = old[co][r] + accx[co][r];
  }
}

// Forward of both per-position MLPs of a hidden-32 layer in one pass (32 x 32 matrices, torch [out][in] layout, pitch 32):
//   out_k = (W1_k LeakyReLU(W0_k x + b0_k) + b1_k) * mask   (model/ppgn.py:58-59),  k = m1, m2 from one read of x.
// One wave per 16-position tile, the hidden tile chained through registers in the MFMA result layout (as mlp32_bwd_kernel): the
// layer-by-layer form (four generic GEMM launches) moves 8 plane-sets for these 3.  hid1 / hid2 (nullable): training keeps the
// hidden activations.
__global__ __launch_bounds__(256) void mlp32_fwd_kernel(const float* __restrict__ X, long x_bs,
                                                        const float* __restrict__ W0a, const float* __restrict__ W0b,
                                                        const float* __restrict__ W1a, const float* __restrict__ W1b,
                                                        const float* __restrict__ b0a, const float* __restrict__ b0b,
                                                        const float* __restrict__ b1a, const float* __restrict__ b1b,
                                                        const float* __restrict__ maskp, float* __restrict__ o1, float* __restrict__ o2,
                                                        float* __restrict__ hid1, float* __restrict__ hid2, long hs, int P, int B) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int ntile = (P + 15) >> 4;
  const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (id >= B * ntile) return;
  const int b = id / ntile, pos = 16 * (id - b * ntile) + n;
  const bool ok = pos < P;
  const size_t g = (size_t)b * hs;
  floatx4 src[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) src[ci][r] = ok ? X[(size_t)b * x_bs + (size_t)(16 * ci + 4 * q + r) * P + pos] : 0.f;
  const float mk = (maskp && ok) ? maskp[(size_t)b * P + pos] : 1.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* W0 = k ? W0b : W0a;
    const float* W1 = k ? W1b : W1a;
    const float* b0 = k ? b0b : b0a;
    const float* b1 = k ? b1b : b1a;
    float* hid = k ? hid2 : hid1;
    float* out = k ? o2 : o1;
    float w0[2][4][2], w1[2][4][2];   // A operand: lane (q, m = n) holds W[16 co + n][16 ci + 4 q + r]
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co) {
          w0[ci][r][co] = W0[(16 * co + n) * 32 + 16 * ci + 4 * q + r];
          w1[ci][r][co] = W1[(16 * co + n) * 32 + 16 * ci + 4 * q + r];
        }
    floatx4 t[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}}, acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co) t[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[ci][r][co], src[ci][r], t[co], 0, 0, 0);
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = t[co][r] + (b0 ? b0[16 * co + 4 * q + r] : 0.f);
        t[co][r] = v > 0.f ? v : kSlope * v;
        if (hid && ok) hid[g + (size_t)(16 * co + 4 * q + r) * P + pos] = t[co][r];
      }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int co = 0; co < 2; ++co) acc[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[ci][r][co], t[ci][r], acc[co], 0, 0, 0);
    if (ok) {
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[g + (size_t)(16 * co + 4 * q + r) * P + pos] = (acc[co][r] + (b1 ? b1[16 * co + 4 * q + r] : 0.f)) * mk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Hidden-32 networks at N > 20, "wide" forms: one wave per 64 consecutive positions, 16-byte accesses.
// v_mfma_f32_16x16x4_f32 as D = W * X: A operand lane (q, n) = W[16 co + n][k], B operand lane (q, n) = X[k][column n], k = 4 q + r
// within a 16-channel block ci.  A column of a tile may be ANY position, so tile j takes the positions 4 n + j of the wave's 64:
// lane (q, n) then loads, per channel 16 ci + 4 q + r, ONE float4 (positions 4 n .. 4 n + 3 = one value for each of the four tiles), a
// row of 16 lanes reads 256 contiguous bytes, and the result lane (q, n) of tile j holds channels 4 q + r of position 4 n + j: again one
// float4 per channel over the four tiles.  The 16-position form above (mlp32_fwd_kernel) moves 64-byte row pieces with dword accesses.
// Results feed the next product from registers in the result layout (the chaining of mlp32_bwd_kernel).
struct W32 { float w[2][4][2]; };                            // [ci][r][co] = W[16 co + n][16 ci + 4 q + r]
__device__ __forceinline__ W32 w32_load(const float* __restrict__ W, int ldw, int q, int n) {
  W32 f;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int co = 0; co < 2; ++co) f.w[ci][r][co] = W[(size_t)(16 * co + n) * ldw + 16 * ci + 4 * q + r];
  return f;
}
struct X32 { float4 v[2][4]; };                              // [ci][r]: channel 16 ci + 4 q + r, positions 4 n .. 4 n + 3
__device__ __forceinline__ X32 x32_load(const float* __restrict__ Xp, int P, int q, int n, bool ok) {   // Xp: channel 0 at the wave's first position
  X32 x;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      x.v[ci][r] = ok ? *reinterpret_cast<const float4*>(Xp + (size_t)(16 * ci + 4 * q + r) * P + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
  return x;
}
__device__ __forceinline__ float f4_at(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
struct A32 { floatx4 a[4][2]; };                             // [tile j][co]: element r = channel 16 co + 4 q + r of position 4 n + j
__device__ __forceinline__ void a32_zero(A32& a) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int co = 0; co < 2; ++co) a.a[j][co] = floatx4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void a32_mma(A32& a, const W32& w, const X32& x) {
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int co = 0; co < 2; ++co) a.a[j][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w[ci][r][co], f4_at(x.v[ci][r], j), a.a[j][co], 0, 0, 0);
}
// the same with the B operand taken from a result in registers (channel block ci of the source = result block co)
__device__ __forceinline__ void a32_mma_chain(A32& a, const W32& w, const A32& t) {
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int co = 0; co < 2; ++co) a.a[j][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w[ci][r][co], t.a[j][ci][r], a.a[j][co], 0, 0, 0);
}
// channel (co, r) of the four tiles as one float4 (positions 4 n .. 4 n + 3)
__device__ __forceinline__ float4 a32_row(const A32& a, int co, int r) { return make_float4(a.a[0][co][r], a.a[1][co][r], a.a[2][co][r], a.a[3][co][r]); }
__device__ __forceinline__ void a32_store(const A32& a, float* __restrict__ Yp, int P, int q, int n, bool ok) {
  if (!ok) return;
#pragma unroll
  for (int co = 0; co < 2; ++co)
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(Yp + (size_t)(16 * co + 4 * q + r) * P + 4 * n) = a32_row(a, co, r);
}
__device__ __forceinline__ void a32_bias(A32& a, const float* __restrict__ b, int q) {
  if (!b) return;
#pragma unroll
  for (int co = 0; co < 2; ++co)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float bv = b[16 * co + 4 * q + r];
#pragma unroll
      for (int j = 0; j < 4; ++j) a.a[j][co][r] += bv;
    }
}
__device__ __forceinline__ void a32_leaky(A32& a) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = a.a[j][co][r]; a.a[j][co][r] = fmaxf(v, kSlope * v); }   // slope < 1: the larger IS the activation (two instructions, not three)
}
// the source's folded instance norm: x * scale + shift per channel; a lane's eight (scale, shift) pairs of the graph's [2][32] table
// (norm_finalize_kernel)
struct T32 { float sc[2][4], sh[2][4]; };
__device__ __forceinline__ T32 t32_load(const float* __restrict__ T, int q) {
  T32 t;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) { t.sc[ci][r] = T[16 * ci + 4 * q + r]; t.sh[ci][r] = T[32 + 16 * ci + 4 * q + r]; }
  return t;
}
__device__ __forceinline__ void x32_affine(X32& x, const T32& t) {
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sc = t.sc[ci][r], sh = t.sh[ci][r];
      float4& v = x.v[ci][r];
      v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
    }
}
__device__ __forceinline__ void x32_mask(X32& x, const float4& mk) {   // exact zeros at masked positions (model/ppgn.py:315)
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float4& v = x.v[ci][r];
      v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
    }
}
// sum over the 16 lanes n of a lane's row (same q), in every lane: four DPP adds (quad swaps, then the mirrored half row and the mirrored
// row bring in the other quad's and the other half's sums); __shfl_xor would be four ds_bpermute round trips through the LDS crossbar
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  return v;
}
// wave -> (graph, first position) of its 64-position group g; P % 4 == 0, a group never crosses a graph
struct G32 { int b, p0; bool ok; };
__device__ __forceinline__ G32 g32_of(int g, int gpp, int P, int n) {
  G32 r;
  r.b = g / gpp; r.p0 = 64 * (g - r.b * gpp); r.ok = r.p0 + 4 * n < P;
  return r;
}

// both per-position MLPs of a hidden-32 layer (model/ppgn.py:58-59), wide form of mlp32_fwd_kernel; persistent waves, both MLPs'
// four matrices resident in registers (64)
__global__ __launch_bounds__(256) void mlp32w_fwd_kernel(const float* __restrict__ X, long x_bs,
                                                         const float* __restrict__ W0a, const float* __restrict__ W0b,
                                                         const float* __restrict__ W1a, const float* __restrict__ W1b,
                                                         const float* __restrict__ b0a, const float* __restrict__ b0b,
                                                         const float* __restrict__ b1a, const float* __restrict__ b1b,
                                                         const float* __restrict__ maskp, float* __restrict__ o1, float* __restrict__ o2,
                                                         float* __restrict__ hid1, float* __restrict__ hid2, long hs, int P, int B,
                                                         const float* __restrict__ aff) {   // aff: null, or [B][2][32] (scale, shift) of the source's instance norm
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int gpp = (P + 63) >> 6, ngroups = B * gpp;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
  const W32 w0a = w32_load(W0a, 32, q, n), w0b = w32_load(W0b, 32, q, n), w1a = w32_load(W1a, 32, q, n), w1b = w32_load(W1b, 32, q, n);
  if (wave0 >= ngroups) return;
  G32 nx = g32_of(wave0, gpp, P, n);
  X32 xn = x32_load(X + (size_t)nx.b * x_bs + nx.p0, P, q, n, nx.ok);
  for (int g = wave0; g < ngroups; g += nwaves) {
    const G32 at = nx;
    X32 x = xn;
    if (g + nwaves < ngroups) {                              // the next group's source is in flight during this group's products
      nx = g32_of(g + nwaves, gpp, P, n);
      xn = x32_load(X + (size_t)nx.b * x_bs + nx.p0, P, q, n, nx.ok);
    }
    // (the table is read where it is used -- L2 hits; carried with the prefetch, 16 more registers, the kernel pair was 9 % slower)
    if (aff) x32_affine(x, t32_load(aff + (size_t)at.b * 64, q));
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (maskp && at.ok) mk = *reinterpret_cast<const float4*>(maskp + (size_t)at.b * P + at.p0 + 4 * n);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      A32 t, acc;
      a32_zero(t);
      a32_mma(t, k ? w0b : w0a, x);
      a32_bias(t, k ? b0b : b0a, q);
      a32_leaky(t);
      float* hid = k ? hid2 : hid1;
      if (hid) a32_store(t, hid + (size_t)at.b * hs + at.p0, P, q, n, at.ok);
      a32_zero(acc);
      a32_mma_chain(acc, k ? w1b : w1a, t);
      a32_bias(acc, k ? b1b : b1a, q);
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc.a[0][co][r] *= mk.x; acc.a[1][co][r] *= mk.y; acc.a[2][co][r] *= mk.z; acc.a[3][co][r] *= mk.w; }
      a32_store(acc, (k ? o2 : o1) + (size_t)at.b * hs + at.p0, P, q, n, at.ok);
    }
  }
}

// The same pair in split-bf16 arithmetic (inference, precision bf16x3): K = 32 is ONE k step of v_mfma_f32_16x16x32_bf16.  With k slot e of
// lane group q standing for channel (e < 4 ? 4 q + e : 16 + 4 q + e - 4), the eight values a lane holds for a position -- loaded float4s,
// or the first product's result in the MFMA layout -- are exactly its B fragment, and the weights are fragmented in that order once.
// Three MFMAs per product (hi hi + hi lo + lo hi, as in the H = 128 kernels): 24 MFMAs of 16 cycles per product and 64 positions in
// place of 64 of 32 cycles.
struct W32S { uintx4 h[2], l[2]; };                          // [co]: A fragment of output block co (hi / lo pieces)
__device__ __forceinline__ W32S w32s_load(const float* __restrict__ W, int ldw, int q, int n) {
  W32S f;
#pragma unroll
  for (int co = 0; co < 2; ++co) {
    const float* row = W + (size_t)(16 * co + n) * ldw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = row[(e < 4 ? 4 * q + e : 16 + 4 * q + e - 4)];
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned h, l = 0; w3_split2<3>(v[2 * e], v[2 * e + 1], h, l); f.h[co][e] = h; f.l[co][e] = l; }
  }
  return f;
}
__device__ __forceinline__ void b32s_split(const float (&v)[8], uintx4& h, uintx4& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { unsigned hh, ll = 0; w3_split2<3>(v[2 * e], v[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
}
__device__ __forceinline__ void a32s_mma(A32& a, const W32S& w, int j, const uintx4& bh, const uintx4& bl) {
#pragma unroll
  for (int co = 0; co < 2; ++co) {
    a.a[j][co] = w3_mfma(w.h[co], bl, a.a[j][co]);
    a.a[j][co] = w3_mfma(w.l[co], bh, a.a[j][co]);
    a.a[j][co] = w3_mfma(w.h[co], bh, a.a[j][co]);
  }
}
__global__ __launch_bounds__(256) void mlp32w3_fwd_kernel(const float* __restrict__ X, long x_bs,
                                                          const float* __restrict__ W0a, const float* __restrict__ W0b,
                                                          const float* __restrict__ W1a, const float* __restrict__ W1b,
                                                          const float* __restrict__ b0a, const float* __restrict__ b0b,
                                                          const float* __restrict__ b1a, const float* __restrict__ b1b,
                                                          const float* __restrict__ maskp, float* __restrict__ o1, float* __restrict__ o2,
                                                          long hs, int P, int B, const float* __restrict__ aff) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int gpp = (P + 63) >> 6, ngroups = B * gpp;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
  const W32S w0a = w32s_load(W0a, 32, q, n), w0b = w32s_load(W0b, 32, q, n), w1a = w32s_load(W1a, 32, q, n), w1b = w32s_load(W1b, 32, q, n);
  if (wave0 >= ngroups) return;
  G32 nx = g32_of(wave0, gpp, P, n);
  X32 xn = x32_load(X + (size_t)nx.b * x_bs + nx.p0, P, q, n, nx.ok);
  for (int g = wave0; g < ngroups; g += nwaves) {
    const G32 at = nx;
    X32 x = xn;
    if (g + nwaves < ngroups) {
      nx = g32_of(g + nwaves, gpp, P, n);
      xn = x32_load(X + (size_t)nx.b * x_bs + nx.p0, P, q, n, nx.ok);
    }
    if (aff) x32_affine(x, t32_load(aff + (size_t)at.b * 64, q));
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (maskp && at.ok) mk = *reinterpret_cast<const float4*>(maskp + (size_t)at.b * P + at.p0 + 4 * n);
    uintx4 xh[4], xl[4];                                     // the source's fragments, one per tile j, shared by both MLPs
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f4_at(x.v[e >> 2][e & 3], j);
      b32s_split(v, xh[j], xl[j]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      A32 t, acc;
      a32_zero(t);
#pragma unroll
      for (int j = 0; j < 4; ++j) a32s_mma(t, k ? w0b : w0a, j, xh[j], xl[j]);
      a32_bias(t, k ? b0b : b0a, q);
      a32_leaky(t);
      a32_zero(acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t.a[j][e >> 2][e & 3];
        uintx4 th, tl;
        b32s_split(v, th, tl);
        a32s_mma(acc, k ? w1b : w1a, j, th, tl);
      }
      a32_bias(acc, k ? b1b : b1a, q);
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc.a[0][co][r] *= mk.x; acc.a[1][co][r] *= mk.y; acc.a[2][co][r] *= mk.z; acc.a[3][co][r] *= mk.w; }
      a32_store(acc, (k ? o2 : o1) + (size_t)at.b * hs + at.p0, P, q, n, at.ok);
    }
  }
}

// m4 of a hidden-32 layer on cat(mm, x) (model/ppgn.py:66-69): Y = W[:, 0:32] mm + W[:, 32:64] x + b, both halves resident (32 registers)
template <bool SPLIT>   // SPLIT: both products as three bf16 MFMAs on split operands (inference, precision bf16x3), as mlp32w3_fwd_kernel
__global__ __launch_bounds__(256) void lin32x2w_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                       const float* __restrict__ X1, long x1_bs, const float* __restrict__ X2, long x2_bs,
                                                       float* __restrict__ Y, long y_bs, int P, int B,
                                                       const float* __restrict__ aff2, const float* __restrict__ maskp, float* __restrict__ stat, float* __restrict__ cnt) {
  // stat != null (instance norm folded, as for H = 128): X2 goes through its own norm (aff2: [B][2][32], null for layer 0), Y is the RAW
  // result with exact zeros at masked positions, and per 64-position group g and channel the masked (mean, centred sum of squares) go to
  // stat[g][32][2], the group's unmasked count to cnt[g] (two passes over the wave's registers; merged by norm_finalize_kernel)
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int gpp = (P + 63) >> 6, ngroups = B * gpp;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
  W32 wa{}, wb{};
  W32S sa{}, sb{};
  if (SPLIT) { sa = w32s_load(W, ldw, q, n); sb = w32s_load(W + 32, ldw, q, n); }
  else { wa = w32_load(W, ldw, q, n); wb = w32_load(W + 32, ldw, q, n); }
  if (wave0 >= ngroups) return;
  G32 nx = g32_of(wave0, gpp, P, n);
  X32 x1n = x32_load(X1 + (size_t)nx.b * x1_bs + nx.p0, P, q, n, nx.ok), x2n = x32_load(X2 + (size_t)nx.b * x2_bs + nx.p0, P, q, n, nx.ok);
  for (int g = wave0; g < ngroups; g += nwaves) {
    const G32 at = nx;
    const X32 x1 = x1n;
    X32 x2 = x2n;
    if (g + nwaves < ngroups) {
      nx = g32_of(g + nwaves, gpp, P, n);
      x1n = x32_load(X1 + (size_t)nx.b * x1_bs + nx.p0, P, q, n, nx.ok);
      x2n = x32_load(X2 + (size_t)nx.b * x2_bs + nx.p0, P, q, n, nx.ok);
    }
    if (aff2) x32_affine(x2, t32_load(aff2 + (size_t)at.b * 64, q));
    A32 acc;
    a32_zero(acc);
    if (SPLIT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
        uintx4 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f4_at(x1.v[e >> 2][e & 3], j);
        b32s_split(v, h, l);
        a32s_mma(acc, sa, j, h, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f4_at(x2.v[e >> 2][e & 3], j);
        b32s_split(v, h, l);
        a32s_mma(acc, sb, j, h, l);
      }
    } else {
      a32_mma(acc, wa, x1);
      a32_mma(acc, wb, x2);
    }
    a32_bias(acc, bias, q);
    if (stat) {
      const float4 mk = at.ok ? *reinterpret_cast<const float4*>(maskp + (size_t)at.b * P + at.p0 + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const bool m0 = mk.x > 0.f, m1 = mk.y > 0.f, m2 = mk.z > 0.f, m3 = mk.w > 0.f;
      const float nt = row16_sum((m0 ? 1.f : 0.f) + (m1 ? 1.f : 0.f) + (m2 ? 1.f : 0.f) + (m3 ? 1.f : 0.f));
      const float rn = 1.f / fmaxf(nt, 1.f);
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = acc.a[0][co][r], v1 = acc.a[1][co][r], v2 = acc.a[2][co][r], v3 = acc.a[3][co][r];
          const float mean = row16_sum(((m0 ? v0 : 0.f) + (m1 ? v1 : 0.f)) + ((m2 ? v2 : 0.f) + (m3 ? v3 : 0.f))) * rn;
          const float d0 = m0 ? v0 - mean : 0.f, d1 = m1 ? v1 - mean : 0.f, d2 = m2 ? v2 - mean : 0.f, d3 = m3 ? v3 - mean : 0.f;
          const float ss = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
          if (n == 0) *reinterpret_cast<float2*>(stat + ((size_t)g * 32 + 16 * co + 4 * q + r) * 2) = make_float2(mean, ss);
          acc.a[0][co][r] = m0 ? v0 : 0.f; acc.a[1][co][r] = m1 ? v1 : 0.f; acc.a[2][co][r] = m2 ? v2 : 0.f; acc.a[3][co][r] = m3 ? v3 : 0.f;
        }
      if (lane == 0) cnt[g] = nt;
    }
    a32_store(acc, Y + (size_t)at.b * y_bs + at.p0, P, q, n, at.ok);
  }
}

// Read-out of a hidden-32 network in one pass (model/ppgn.py:318-333): R = layer_cat_lin(cat([A, sigma I, u_1 .. u_L])),
// act = LeakyReLU(after_conv(R)), logits = final_lin(R + act).  The L + 1 weight blocks of layer_cat_lin go through LDS ([k][33]
// floats: the A-operand reads of 16 consecutive rows are conflict-free), after_conv / final_lin stay in registers; R and act never
// leave the wave unless training asks for them (Rs / As, nullable).  logits [B][N][N]; ld = padded row length of the planes.
constexpr int H32_MAXL = 16;
// SPLIT (inference, precision bf16x3): layer_cat_lin's blocks and after_conv as three bf16 MFMAs per product on split operands (see
// mlp32w3_fwd_kernel); the blocks' A fragments are split once per workgroup into LDS behind the float table ([L][2 co][hi, lo][64 lanes] uintx4)
template <bool SPLIT>
__global__ __launch_bounds__(256) void head32w_kernel(const float* __restrict__ cat, long cat_bs, const float* __restrict__ Wc, int ldc, const float* __restrict__ bc,
                                                      const float* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ wf, const float* __restrict__ bf,
                                                      float* __restrict__ Rs, float* __restrict__ As, long hs, float* __restrict__ logits, int N, int ld, int L, int B,
                                                      const float* __restrict__ aff, const float* __restrict__ maskp) {   // aff: null, or [L][B][2][32]: the sources are RAW (with maskp)
  extern __shared__ float wlds[];                            // [(2 + 32 L)][33]
  const int P = N * ld, K = 2 + 32 * L;
  for (int i = threadIdx.x; i < 32 * K; i += blockDim.x) { const int o = i / K, k = i - o * K; wlds[k * 33 + o] = Wc[(size_t)o * ldc + k]; }
  uintx4* frag = reinterpret_cast<uintx4*>(wlds + ((K * 33 + 3) & ~3));
  if (SPLIT) {
    for (int i = threadIdx.x; i < L * 2 * 64; i += blockDim.x) {
      const int ln = i & 63, co = (i >> 6) & 1, l = i >> 7, fq = ln >> 4, fn = ln & 15;
      const float* row = Wc + (size_t)(16 * co + fn) * ldc + 2 + 32 * l;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = row[(e < 4 ? 4 * fq + e : 16 + 4 * fq + e - 4)];
      uintx4 h, lo;
      b32s_split(v, h, lo);
      frag[((l * 2 + co) * 2 + 0) * 64 + ln] = h;
      frag[((l * 2 + co) * 2 + 1) * 64 + ln] = lo;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int gpp = (P + 63) >> 6, ngroups = B * gpp;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
  W32 wac{};
  W32S wacs{};
  if (SPLIT) wacs = w32s_load(Wa, 32, q, n); else wac = w32_load(Wa, 32, q, n);
  float wfin[2][4];
#pragma unroll
  for (int co = 0; co < 2; ++co)
#pragma unroll
    for (int r = 0; r < 4; ++r) wfin[co][r] = wf[16 * co + 4 * q + r];
  const float bfin = bf[0];
  for (int g = wave0; g < ngroups; g += nwaves) {
    const G32 at = g32_of(g, gpp, P, n);
    const float* cp = cat + (size_t)at.b * cat_bs + at.p0;
    A32 R;
    a32_zero(R);
    {   // the two input planes: rank-2 update
      const float4 a4 = at.ok ? *reinterpret_cast<const float4*>(cp + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 s4 = at.ok ? *reinterpret_cast<const float4*>(cp + P + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * co + 4 * q + r;
          const float c0 = wlds[0 * 33 + o], c1 = wlds[1 * 33 + o];
#pragma unroll
          for (int j = 0; j < 4; ++j) R.a[j][co][r] = c0 * f4_at(a4, j) + c1 * f4_at(s4, j);
        }
    }
    float4 mk = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aff && at.ok) mk = *reinterpret_cast<const float4*>(maskp + (size_t)at.b * P + at.p0 + 4 * n);
    for (int l = 0; l < L; ++l) {
      X32 x = x32_load(cp + (size_t)(2 + 32 * l) * P, P, q, n, at.ok);
      if (aff) { x32_affine(x, t32_load(aff + ((size_t)l * B + at.b) * 64, q)); x32_mask(x, mk); }
      if (SPLIT) {
        W32S ws;
#pragma unroll
        for (int co = 0; co < 2; ++co) { ws.h[co] = frag[((l * 2 + co) * 2 + 0) * 64 + lane]; ws.l[co] = frag[((l * 2 + co) * 2 + 1) * 64 + lane]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[8];
          uintx4 h, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = f4_at(x.v[e >> 2][e & 3], j);
          b32s_split(v, h, lo);
          a32s_mma(R, ws, j, h, lo);
        }
      } else {
        W32 w;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int co = 0; co < 2; ++co) w.w[ci][r][co] = wlds[(2 + 32 * l + 16 * ci + 4 * q + r) * 33 + 16 * co + n];
        a32_mma(R, w, x);
      }
    }
    a32_bias(R, bc, q);
    if (Rs) a32_store(R, Rs + (size_t)at.b * hs + at.p0, P, q, n, at.ok);
    A32 act;
    a32_zero(act);
    if (SPLIT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
        uintx4 h, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = R.a[j][e >> 2][e & 3];
        b32s_split(v, h, lo);
        a32s_mma(act, wacs, j, h, lo);
      }
    } else {
      a32_mma_chain(act, wac, R);
    }
    a32_bias(act, ba, q);
    a32_leaky(act);
    if (As) a32_store(act, As + (size_t)at.b * hs + at.p0, P, q, n, at.ok);
    float part[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sacc = 0.f;
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc += wfin[co][r] * (R.a[j][co][r] + act.a[j][co][r]);
      sacc += __shfl_xor(sacc, 16, 64);
      sacc += __shfl_xor(sacc, 32, 64);
      part[j] = sacc + bfin;
    }
    if (q == 0 && at.ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = at.p0 + 4 * n + j, i = p / ld, jj = p - i * ld;
        if (jj < N) logits[((size_t)at.b * N + i) * N + jj] = part[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------------------
struct Ctx {
  hipStream_t st;
  Layout w;
  float* ws;
  float* at(size_t off) const { return ws + off; }
};

struct Geo { int B, P; hipStream_t st; };

// in_lin (model/ppgn.py:301): u0[b][c][p] = w[c][0] * A[b][p] + w[c][1] * sigmaI[b][p] + bias[c].  K = 2: pure streaming, one block
// reads 1024 positions of the two input planes once and writes them for every channel.
__global__ __launch_bounds__(256) void inlin_kernel(const float* __restrict__ cat, long cat_bs, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int H, int P, float* __restrict__ u0) {
  const int b = blockIdx.y;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= P) return;
  const float4 a = *reinterpret_cast<const float4*>(cat + (size_t)b * cat_bs + p);
  const float4 s = *reinterpret_cast<const float4*>(cat + (size_t)b * cat_bs + P + p);
  float* out = u0 + (size_t)b * H * P + p;
  for (int c = 0; c < H; ++c) {
    const float w0 = w[2 * c], w1 = w[2 * c + 1], bc = bias[c];
    float4 v;
    v.x = ws_inlin(w0, w1, bc, a.x, s.x); v.y = ws_inlin(w0, w1, bc, a.y, s.y); v.z = ws_inlin(w0, w1, bc, a.z, s.z); v.w = ws_inlin(w0, w1, bc, a.w, s.w);
    *reinterpret_cast<float4*>(out + (size_t)c * P) = v;
  }
}

// A/B switch for measurements: DPPM_DISABLE_WSTAT=1 routes the H = 128 linears through the generic GEMM family.
inline bool use_wstat(int H) {
  const bool off = sw(SW_DISABLE_WSTAT) != 0;
  return H == WS_H && !off;
}

// weight-stationary Linear in the requested arithmetic (include/dppm_hip.h DPPM_PREC_*)
inline int ws_lin_p(const WsArgs& a, int nsrc, bool trans, int prec, hipStream_t st) {
  // training (DPPM_PREC_F32X6): single-source linears on the bf16 pipe with the three-piece split (dppm_w6.hip); the two-source form
  // does not fit (two matrices x three pieces = 96 registers beside accumulators and fragments) and stays exact f32
  if (prec == DPPM_PREC_F32X6 && nsrc == 1 && sw(SW_DISABLE_W6) == 0 && w6_lin_ok(a, nsrc)) return w6_lin_run(a, nsrc, trans, st);
  if (prec == DPPM_PREC_BF16X3) return w3_lin<3>(a, nsrc, trans, st);
  if (prec == DPPM_PREC_BF16) return w3_lin<1>(a, nsrc, trans, st);
  return ws_lin(a, nsrc, trans, st);
}
inline bool valid_prec(int prec) { return prec == DPPM_PREC_F32 || prec == DPPM_PREC_BF16X3 || prec == DPPM_PREC_BF16 || prec == DPPM_PREC_F32X6; }

GemmArgs base_args() {
  GemmArgs a{};
  a.nseg = 1; a.nsplit = 1; a.kchunk = 0; a.scale_div = 1; a.mul_div = 1;
  return a;
}

// the wide hidden-32 kernels apply: 16-byte aligned planes whose rows are multiples of four positions
inline bool wide32_ok(int P, const float* x, long x_bs, const float* y, long y_bs) {
  return sw(SW_DISABLE_WIDE32) == 0 && P % 4 == 0 && aligned16(x) && aligned16(y) && x_bs % 4 == 0 && y_bs % 4 == 0;
}
inline int wide32_grid(int B, int P) {   // persistent waves: four per block, up to 8 blocks per CU's worth of work in flight
  const long groups = (long)B * ((P + 63) / 64);
  const long blocks = (groups + 3) / 4;
  return (int)(blocks < 2048 ? blocks : 2048);
}

// Y[b] (M x P) = epilogue( sum_s W_s (M x K_s) * X_s[b] (K_s x P) )
struct LinArgs {
  const float* W; int ldw; bool w_trans;   // w_trans: use W^T (W is stored [K][M] with pitch ldw)
  const float* X; long x_bs; int K;
};
int linear(const Geo& c, int M, const LinArgs& s0, const LinArgs* s1, const float* bias, int act, const float* mulmask,
           const float* dact, long dact_bs, float* Y, long y_bs, bool accumulate);
int linear(const Ctx& c, int M, const LinArgs& s0, const LinArgs* s1, const float* bias, int act, const float* mulmask,
           const float* dact, long dact_bs, float* Y, long y_bs, bool accumulate) {
  return linear(Geo{c.w.B, c.w.P, c.st}, M, s0, s1, bias, act, mulmask, dact, dact_bs, Y, y_bs, accumulate);
}
int linear(const Geo& c, int M, const LinArgs& s0, const LinArgs* s1, const float* bias, int act, const float* mulmask,
           const float* dact, long dact_bs, float* Y, long y_bs, bool accumulate) {
  const bool l32_off = sw(SW_DISABLE_LIN32) != 0;   // A/B switch for measurements
  if (M == 32 && s0.K == 32 && s0.w_trans && !s1 && !bias && !act && !mulmask && !l32_off) {
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    hipLaunchKernelGGL(lin32t_kernel, dim3(cdiv((long)c.B * cdiv(c.P, 16), 4)), dim3(256), 0, c.st, s0.W, s0.ldw, s0.X, s0.x_bs, dact, dact_bs,
                       Y, y_bs, c.P, c.B, accumulate ? 1 : 0, (const float*)nullptr, (float*)nullptr, 0L, 0);
    if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 2.0 * 32 * 32 * (double)c.P * c.B);
    return check_launch("lin32t");
  }
  GemmArgs a = base_args();
  const LinArgs* segs[2] = {&s0, s1};
  a.nseg = s1 ? 2 : 1;
  a.a_vec = 1;
  for (int i = 0; i < a.nseg; ++i) {
    a.seg[i].A = segs[i]->W; a.seg[i].a_bs = 0; a.seg[i].lda = segs[i]->ldw;
    a.seg[i].B = segs[i]->X; a.seg[i].b_bs = segs[i]->x_bs; a.seg[i].ldb = c.P;
    a.seg[i].K = segs[i]->K;
    if (segs[i]->w_trans != s0.w_trans) return fail(DPPM_E_INVALID, "linear: segments must share the weight orientation");
    if (!aligned16(segs[i]->W) || segs[i]->ldw % 4 != 0) a.a_vec = 0;
  }
  a.b_vec = 1;   // planes are 16-byte aligned by construction (P % 4 == 0, workspace 256-byte aligned)
  a.M = M; a.N = c.P;
  a.C = Y; a.c_bs = y_bs; a.ldc = c.P;
  a.bias = bias; a.act = act;
  a.mul = mulmask; a.mul_bs = c.P; a.mul_ld = 0; a.mul_div = 1;
  a.dact = dact; a.dact_bs = dact_bs;
  a.accumulate = accumulate ? 1 : 0;
  a.prof_class = PROF_LINEAR;
  return launch_gemm(a, c.B, /*a_kmajor=*/s0.w_trans, /*b_kmajor=*/true, c.st);
}

// m1, m2: Linear -> LeakyReLU -> Linear, then * mask   (model/ppgn.py:58-59)
struct InlSrc { const float* w; const float* b; const float* in; long bs; };   // in_lin folded into the consumer (WsArgs::inl_*)
int mlp_pair(const Geo& c, const dppm_ppgn_layer& q, int H, const float* x, long x_bs, const float* maskp, float* o1, float* o2,
             float* hid1, float* hid2, bool save_hid, int prec, const float* aff = nullptr, const InlSrc* inl = nullptr) {
  const long hs = (long)H * c.P;
  if (use_wstat(H)) {   // weight-stationary fused pair: hidden activations stay in LDS unless training
    WsArgs wa{};
    wa.njobs = 2; wa.maskp = maskp; wa.B = c.B; wa.P = c.P;
    wa.aff = aff;   // split inference forms only (checked by the caller)
    if (inl) { wa.inl_w = inl->w; wa.inl_b = inl->b; wa.inl_in = inl->in; wa.inl_bs = inl->bs; }
    const float* w0[2] = {q.m1_w0, q.m2_w0}; const float* w1[2] = {q.m1_w1, q.m2_w1};
    const float* b0[2] = {q.m1_b0, q.m2_b0}; const float* b1[2] = {q.m1_b1, q.m2_b1};
    float* out[2] = {o1, o2}; float* hid[2] = {hid1, hid2};
    for (int k = 0; k < 2; ++k) {
      WsJob& j = wa.job[k];
      j.X = x; j.x_bs = x_bs; j.W1 = w0[k]; j.W2 = w1[k]; j.ldw1 = H; j.ldw2 = H; j.b1 = b0[k]; j.b2 = b1[k];
      j.Mid = save_hid ? hid[k] : nullptr; j.mid_bs = hs; j.Y = out[k]; j.y_bs = hs;
    }
    // split inference modes: the role-split form (dppm_w4.hip; 0.54 ms per pair at N = 200, B = 32) unless DPPM_W3MLP=1 asks for the
    // two-group form of dppm_wsplit.h (0.57 ms), which cannot take the in_lin source and needs per-graph norm tables that fit its LDS
    if (!save_hid && prec != DPPM_PREC_F32 && sw(SW_DISABLE_W4) == 0 && w4_mlp_ok(wa)) {
      const int tpp = (c.P + WS_TP - 1) / WS_TP;
      const bool tables_fit = w3_contig_ok((long)c.B * tpp, tpp, (int)(8 * w3_slots(2, (long)c.B * tpp)));
      if ((aff && !tables_fit) || inl || sw(SW_W3MLP) == 0) return w4_mlp_run(wa, prec == DPPM_PREC_BF16X3 ? 3 : 1, c.st);
    }
    if (inl) return fail(DPPM_E_INVALID, "mlp_pair: the in_lin form needs the role-split kernel");
    // (Training forward: the fused exact-f32 pair below stays.  Two launches of the three-piece-split Linear would run at the HBM
    // roof -- 0.95 ms against 1.45 ms -- and come out CLOSER to float64 in the logits (1.1x torch-f32's distance against 1.75x), but
    // on the ill-conditioned network of test_backward_kinkfree_h128 the gradients then land 5-6.5x torch-f32's distance from float64
    // instead of 3.3x, six- or eight-product form alike: the 4x bar is kept, the backward below takes the split alone.)
    if (!save_hid && prec == DPPM_PREC_BF16X3) return w3_mlp_forward<3>(wa, c.st);
    if (!save_hid && prec == DPPM_PREC_BF16) return w3_mlp_forward<1>(wa, c.st);
    return ws_mlp_forward(wa, save_hid, c.st);   // the saved hidden activations come from the exact-f32 kernel
  }
  const bool mlp32_off = sw(SW_DISABLE_MLP32) != 0;   // A/B switch for measurements
  if (H == 32 && !mlp32_off && wide32_ok(c.P, x, x_bs, o1, hs) && aligned16(o2) && (!maskp || aligned16(maskp)) &&
      (!save_hid || (aligned16(hid1) && aligned16(hid2)))) {   // 64 positions per wave, 16-byte accesses
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    if (!save_hid && prec == DPPM_PREC_BF16X3 && sw(SW_DISABLE_MLP32S) == 0) {   // inference in the split arithmetic
      hipLaunchKernelGGL(mlp32w3_fwd_kernel, dim3(wide32_grid(c.B, c.P)), dim3(256), 0, c.st, x, x_bs, (const float*)q.m1_w0, (const float*)q.m2_w0,
                         (const float*)q.m1_w1, (const float*)q.m2_w1, (const float*)q.m1_b0, (const float*)q.m2_b0, (const float*)q.m1_b1, (const float*)q.m2_b1,
                         maskp, o1, o2, hs, c.P, c.B, aff);
      if (prof) prof_after(c.st, PROF_LINEAR, AR_BF16X3, 8.0 * 32 * 32 * (double)c.P * c.B, 3.0 * sizeof(float) * 32 * (double)c.P * c.B);
      return check_launch("mlp32w3_fwd");
    }
    hipLaunchKernelGGL(mlp32w_fwd_kernel, dim3(wide32_grid(c.B, c.P)), dim3(256), 0, c.st, x, x_bs, (const float*)q.m1_w0, (const float*)q.m2_w0,
                       (const float*)q.m1_w1, (const float*)q.m2_w1, (const float*)q.m1_b0, (const float*)q.m2_b0, (const float*)q.m1_b1, (const float*)q.m2_b1,
                       maskp, o1, o2, save_hid ? hid1 : (float*)nullptr, save_hid ? hid2 : (float*)nullptr, hs, c.P, c.B, aff);
    if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 8.0 * 32 * 32 * (double)c.P * c.B, (save_hid ? 5.0 : 3.0) * sizeof(float) * 32 * (double)c.P * c.B);
    return check_launch("mlp32w_fwd");
  }
  if (aff) return fail(DPPM_E_INVALID, "mlp_pair: a folded instance norm needs the weight-stationary kernels");
  if (H == 32 && !mlp32_off) {   // hidden-32 networks: both MLPs of the layer in one streaming launch
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    hipLaunchKernelGGL(mlp32_fwd_kernel, dim3(cdiv((long)c.B * cdiv(c.P, 16), 4)), dim3(256), 0, c.st, x, x_bs, (const float*)q.m1_w0, (const float*)q.m2_w0,
                       (const float*)q.m1_w1, (const float*)q.m2_w1, (const float*)q.m1_b0, (const float*)q.m2_b0, (const float*)q.m1_b1, (const float*)q.m2_b1,
                       maskp, o1, o2, save_hid ? hid1 : (float*)nullptr, save_hid ? hid2 : (float*)nullptr, hs, c.P, c.B);
    if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 8.0 * 32 * 32 * (double)c.P * c.B, (save_hid ? 5.0 : 3.0) * sizeof(float) * 32 * (double)c.P * c.B);
    return check_launch("mlp32_fwd");
  }
  DPPM_TRY(linear(c, H, LinArgs{q.m1_w0, H, false, x, x_bs, H}, nullptr, q.m1_b0, 1, nullptr, nullptr, 0, hid1, hs, false));
  DPPM_TRY(linear(c, H, LinArgs{q.m1_w1, H, false, hid1, hs, H}, nullptr, q.m1_b1, 0, maskp, nullptr, 0, o1, hs, false));
  DPPM_TRY(linear(c, H, LinArgs{q.m2_w0, H, false, x, x_bs, H}, nullptr, q.m2_b0, 1, nullptr, nullptr, 0, hid2, hs, false));
  return linear(c, H, LinArgs{q.m2_w1, H, false, hid2, hs, H}, nullptr, q.m2_b1, 0, maskp, nullptr, 0, o2, hs, false);
}

// dW (M x Nw, pitch ldo) = sum_b dY[b] (M x P) * X[b]^T (P x Nw) ; optional db = row sums of dY
int wgrad_raw(const Geo& c, int M, const float* dY, long dy_bs, int Nw, const float* X, long x_bs, float* dW, int ldo,
              float* db, float* slabs, size_t slab_floats, float* rsum) {
  const int nsplit = wgrad_nsplit(c.B, c.P), kchunk = wgrad_kchunk(c.B, c.P);
  const int nz = c.B * nsplit;
  if ((size_t)nz * M * Nw > slab_floats) return fail(DPPM_E_INVALID, "wgrad: slab buffer too small");
  const bool w32_off = sw(SW_DISABLE_WGRAD32) != 0;   // A/B switch for measurements
  if (M == 32 && Nw == 32 && kchunk <= W32_MAXK && !w32_off && aligned16(dY) && aligned16(X) && dy_bs % 4 == 0 && x_bs % 4 == 0 && c.P % 4 == 0) {
    const size_t lds = 2 * 32 * (size_t)(kchunk + 4) * sizeof(float);
    DPPM_TRY(lds_limit(reinterpret_cast<const void*>(&wgrad32_kernel), (size_t)((2 * 32 * (W32_MAXK + 4) * sizeof(float))), "wgrad32"));
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    hipLaunchKernelGGL(wgrad32_kernel, dim3(nz), dim3(256), lds, c.st, dY, dy_bs, X, x_bs, c.P, nsplit, kchunk, slabs, db ? rsum : (float*)nullptr);
    if (prof) prof_after(c.st, PROF_WGRAD, AR_F32, 2.0 * 32 * 32 * (double)c.P * c.B);
    DPPM_TRY(check_launch("wgrad32"));
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv((long)M * Nw, RS_E) + (db ? cdiv(M, RS_E) : 0)), dim3(RS_E * RS_G), 0, c.st,
                       (const float*)slabs, nz, M, Nw, dW, ldo, (const float*)rsum, db);
    return check_launch("reduce_slabs(dW, db)");
  }
  GemmArgs a = base_args();
  a.seg[0].A = dY; a.seg[0].a_bs = dy_bs; a.seg[0].lda = c.P;
  a.seg[0].B = X; a.seg[0].b_bs = x_bs; a.seg[0].ldb = c.P;
  a.seg[0].K = c.P;
  a.a_vec = aligned16(dY) && dy_bs % 4 == 0 && c.P % 4 == 0;
  a.b_vec = aligned16(X) && x_bs % 4 == 0 && c.P % 4 == 0;
  a.nsplit = nsplit; a.kchunk = kchunk;
  a.M = M; a.N = Nw;
  a.C = slabs; a.c_bs = (long)M * Nw; a.ldc = Nw;
  a.rowsum = db ? rsum : nullptr;
  a.prof_class = PROF_WGRAD;
  DPPM_TRY(launch_gemm(a, nz, false, false, c.st));
  // dW and (when asked for) db in one launch
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv((long)M * Nw, RS_E) + (db ? cdiv(M, RS_E) : 0)), dim3(RS_E * RS_G), 0, c.st,
                     (const float*)slabs, nz, M, Nw, dW, ldo, (const float*)rsum, db);
  return check_launch("reduce_slabs(dW, db)");
}
int wgrad(const Ctx& c, int M, const float* dY, long dy_bs, int Nw, const float* X, long x_bs, float* dW, int ldo, float* db) {
  if (use_wstat(M) && Nw == WS_H && c.w.slab_floats >= ws_wgrad_scratch_floats()) {   // register-resident split-K form (dppm_wgrad.h)
    // three-piece split products on the bf16 pipe (float32-grade, dppm_wgrad.h); DPPM_WGRAD_F32=1 = exact-f32 MFMAs (A/B switch)
    const bool split = sw(SW_WGRAD_F32) == 0;
    return ws_wgrad(dY, dy_bs, X, x_bs, c.w.B, c.w.P, dW, ldo, db, c.at(c.w.slabs), c.st, split);
  }
  return wgrad_raw(Geo{c.w.B, c.w.P, c.st}, M, dY, dy_bs, Nw, X, x_bs, dW, ldo, db, c.at(c.w.slabs), c.w.slab_floats, c.at(c.w.rsum));
}

// several 32 x 32 weight gradients of the small networks in two launches (wgrad32m_kernel + reduce32m_kernel)
bool wgrad32_multi_ok(const Ctx& c) {
  const bool off = sw(SW_DISABLE_WGRAD32) != 0;
  return !off && c.w.H == 32 && c.w.training && c.w.P % 4 == 0 && wgrad_kchunk(c.w.B, c.w.P) <= W32_MAXK;
}
// `pend` != nullptr (a backward pass over a small network, W32_DEFER_*): the launch writes into its own set of slabs and the fold is
// left to wgrad32_flush, one launch for every layer's jobs; otherwise the fold follows at once.
struct W32Pending { R32Args r; };
int wgrad32_flush(const Ctx& c, W32Pending& pd) {
  if (pd.r.njobs == 0) return DPPM_OK;
  hipLaunchKernelGGL(reduce32m_kernel, dim3(pd.r.njobs * 33), dim3(RS_E * RS_G), 0, c.st, pd.r);
  pd.r.njobs = 0;
  return check_launch("reduce32m");
}
int wgrad32_multi(const Ctx& c, const W32Job* jobs, int njobs, W32Pending* pend = nullptr) {
  W32Args a{};
  a.njobs = njobs; a.P = c.w.P; a.nsplit = wgrad_nsplit(c.w.B, c.w.P); a.kchunk = wgrad_kchunk(c.w.B, c.w.P); a.nz = c.w.B * a.nsplit;
  DPPM_REQUIRE(njobs >= 1 && njobs <= W32_MAXJOBS, "wgrad32_multi: %d jobs", njobs);
  const bool defer = pend && c.w.w32_jobs > W32_MAXJOBS && pend->r.njobs + njobs <= c.w.w32_jobs && pend->r.njobs + njobs <= R32_MAXJOBS;
  float* slabs0 = c.at(c.w.w32);
  float* rsum0 = slabs0 + (size_t)c.w.w32_jobs * a.nz * 1024;
  const int jbase = defer ? pend->r.njobs : 0;   // (deferred sets are packed by job: set s starts where set s - 1 ended)
  a.slabs = slabs0 + (size_t)jbase * a.nz * 1024; a.rsum = rsum0 + (size_t)jbase * a.nz * 32;
  for (int j = 0; j < njobs; ++j) {
    a.job[j] = jobs[j];
    DPPM_REQUIRE(aligned16(jobs[j].dY) && aligned16(jobs[j].X) && jobs[j].dy_bs % 4 == 0 && jobs[j].x_bs % 4 == 0, "wgrad32_multi: unaligned operand");
  }
  const size_t lds = 2 * 32 * (size_t)(a.kchunk + 4) * sizeof(float);
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(&wgrad32m_kernel), (size_t)((2 * 32 * (W32_MAXK + 4) * sizeof(float))), "wgrad32_multi"));
  const bool prof = prof_enabled();
  if (prof) prof_before(c.st);
  hipLaunchKernelGGL(wgrad32m_kernel, dim3(njobs * a.nz), dim3(256), lds, c.st, a);
  if (prof) prof_after(c.st, PROF_WGRAD, AR_F32, 2.0 * 32 * 32 * (double)c.w.P * c.w.B * njobs);
  DPPM_TRY(check_launch("wgrad32m"));
  if (defer) {
    pend->r.nz = a.nz; pend->r.slabs = slabs0; pend->r.rsum = rsum0;
    for (int j = 0; j < njobs; ++j) pend->r.job[pend->r.njobs++] = R32Out{jobs[j].dW, jobs[j].ldo, jobs[j].db, jobs[j].ncols};
    return DPPM_OK;
  }
  if (pend) DPPM_TRY(wgrad32_flush(c, *pend));   // (out of sets: fold what is pending before the shared region would be reused)
  R32Args r{};
  r.njobs = njobs; r.nz = a.nz; r.slabs = a.slabs; r.rsum = a.rsum;
  for (int j = 0; j < njobs; ++j) r.job[j] = R32Out{jobs[j].dW, jobs[j].ldo, jobs[j].db, jobs[j].ncols};
  hipLaunchKernelGGL(reduce32m_kernel, dim3(njobs * 33), dim3(RS_E * RS_G), 0, c.st, r);
  return check_launch("reduce32m");
}

// per-plane products for all (b,c): C = op(A) * op(B) * inv_norm[b] (* mask[b])
int plane_mm(hipStream_t st, const float* A, const float* Bm, float* C, int batch, int n, int ld, bool ta, bool tb,
             const float* scale, const float* mask, int div, int prec = DPPM_PREC_F32) {
  const bool strip_off = sw(SW_DISABLE_STRIPMM) != 0;   // A/B switch for measurements
  if ((prec == DPPM_PREC_BF16X3 || prec == DPPM_PREC_BF16) && !ta && !tb && plane3_supported(n) && aligned16(A) && aligned16(Bm) && aligned16(C)) {
    PlaneArgs pa{A, Bm, C, n, ld, scale, mask, div};   // forward form on the bf16 pipe (dppm_plane4.hip, dppm_planesplit.h)
    const int ns = prec == DPPM_PREC_BF16X3 ? 3 : 1;
    if (sw(SW_DISABLE_PLANE4) == 0 && plane4_ok(pa, batch, ns, false, false)) return plane4_run(pa, batch, ns, false, false, st);
    return prec == DPPM_PREC_BF16X3 ? plane3_launch<3>(pa, batch, st) : plane3_launch<1>(pa, batch, st);
  }
  if (prec == DPPM_PREC_F32X6 && aligned16(A) && aligned16(Bm) && aligned16(C) && sw(SW_DISABLE_PLANE6) == 0) {
    PlaneArgs pa{A, Bm, C, n, ld, scale, mask, div};   // float32-equivalent three-piece split on the bf16 pipe, all three forms (dppm_plane4.hip)
    if (plane4_ok(pa, batch, 6, ta, tb)) return plane4_run(pa, batch, 6, ta, tb, st);
  }
  if (plane_mm_supported(n) && !strip_off && aligned16(A) && aligned16(Bm) && aligned16(C)) {
    PlaneArgs pa{A, Bm, C, n, ld, scale, mask, div};
    return plane_mm_launch(pa, batch, /*a_kmajor=*/ta, /*b_kmajor=*/!tb, st);
  }
  GemmArgs a = base_args();
  const long P = (long)n * ld;
  a.seg[0].A = A; a.seg[0].a_bs = P; a.seg[0].lda = ld;
  a.seg[0].B = Bm; a.seg[0].b_bs = P; a.seg[0].ldb = ld;
  a.seg[0].K = n;
  a.a_vec = aligned16(A); a.b_vec = aligned16(Bm);
  a.M = n; a.N = n; a.n_store = ld;   // pad columns of the output plane are zeroed
  a.C = C; a.c_bs = P; a.ldc = ld;
  a.scale = scale; a.scale_div = div;
  a.mul = mask; a.mul_bs = P; a.mul_ld = ld; a.mul_div = div;
  // op(A) = A   : element (m,k) at A[m*ld + k]  -> K-minor ;  op(A) = A^T : (m,k) at A[k*ld + m] -> K-major
  // op(B) = B   : element (k,n) at B[k*ld + n]  -> K-major ;  op(B) = B^T : (k,n) at B[n*ld + k] -> K-minor
  const bool a_kmajor = ta, b_kmajor = !tb;
  a.prof_class = PROF_PLANE_MM;
  return launch_gemm(a, batch, a_kmajor, b_kmajor, st);
}

int check_params(const dppm_ppgn_params* p, const char* who) {
  DPPM_REQUIRE(p, "%s: null params", who);
  DPPM_REQUIRE(p->num_layers >= 1 && p->num_layers <= 64, "%s: num_layers=%d", who, p->num_layers);
  DPPM_REQUIRE(p->hidden >= 4 && p->hidden % 4 == 0, "%s: hidden=%d must be a positive multiple of 4", who, p->hidden);
  DPPM_REQUIRE(p->in_w && p->in_b && p->cat_w && p->cat_b && p->after_w && p->after_b && p->final_w && p->final_b && p->layers_host,
               "%s: null parameter pointer", who);
  for (int l = 0; l < p->num_layers; ++l) {
    const dppm_ppgn_layer& q = p->layers_host[l];
    DPPM_REQUIRE(q.m1_w0 && q.m1_b0 && q.m1_w1 && q.m1_b1 && q.m2_w0 && q.m2_b0 && q.m2_w1 && q.m2_b1 && q.m4_w && q.m4_b,
                 "%s: null parameter pointer in layer %d", who, l);
  }
  return DPPM_OK;
}

}  // namespace
}  // namespace dppm

using namespace dppm;

extern "C" size_t dppm_ppgn_workspace_bytes(int B, int N, int H, int L, int training) {
  if (B <= 0 || N <= 0 || H <= 0 || L <= 0) return 0;
  return make_layout(B, N, H, L, training).total * sizeof(float);
}

extern "C" int dppm_plane_matmul(const float* A, const float* Bm, float* C, int batch, int n, int ld, int trans_a, int trans_b,
                                 const float* scale, const float* mask, int scale_div, int precision, void* stream) {
  DPPM_REQUIRE(A && Bm && C, "dppm_plane_matmul: null pointer");
  DPPM_REQUIRE(batch > 0 && n > 0 && ld >= n && ld % 4 == 0, "dppm_plane_matmul: batch=%d n=%d ld=%d", batch, n, ld);
  DPPM_REQUIRE(!(trans_a && trans_b), "dppm_plane_matmul: A^T * B^T is not provided");
  DPPM_REQUIRE(scale_div >= 1, "dppm_plane_matmul: scale_div=%d", scale_div);
  DPPM_REQUIRE(valid_prec(precision), "dppm_plane_matmul: precision=%d", precision);
  return plane_mm(as_stream(stream), A, Bm, C, batch, n, ld, trans_a != 0, trans_b != 0, scale, mask, scale_div, precision);
}

extern "C" int dppm_linear_fwd(const float* W, int ldw, int trans_w, const float* X, long x_bs, int K, const float* bias, int act,
                               const float* maskp, float* Y, long y_bs, int M, int B, int P, int accumulate, void* stream) {
  DPPM_REQUIRE(W && X && Y, "dppm_linear_fwd: null pointer");
  DPPM_REQUIRE(M > 0 && K > 0 && B > 0 && P > 0 && P % 4 == 0, "dppm_linear_fwd: M=%d K=%d B=%d P=%d (P must be a multiple of 4)", M, K, B, P);
  DPPM_REQUIRE(ldw >= (trans_w ? M : K), "dppm_linear_fwd: ldw=%d too small", ldw);
  DPPM_REQUIRE(aligned16(X) && aligned16(Y) && x_bs % 4 == 0 && y_bs % 4 == 0, "dppm_linear_fwd: planes must be 16-byte aligned");
  return linear(Geo{B, P, as_stream(stream)}, M, LinArgs{W, ldw, trans_w != 0, X, x_bs, K}, nullptr, bias, act ? 1 : 0, maskp, nullptr, 0,
                Y, y_bs, accumulate != 0);
}

extern "C" int dppm_mlp_pair_fwd(const float* X, long x_bs, const dppm_ppgn_layer* q, int H, const float* maskp, float* o1, float* o2,
                                 float* hid1, float* hid2, int B, int P, int precision, void* stream) {
  DPPM_REQUIRE(X && q && o1 && o2, "dppm_mlp_pair_fwd: null pointer");
  DPPM_REQUIRE(q->m1_w0 && q->m1_b0 && q->m1_w1 && q->m1_b1 && q->m2_w0 && q->m2_b0 && q->m2_w1 && q->m2_b1, "dppm_mlp_pair_fwd: null weight");
  DPPM_REQUIRE(H >= 4 && H % 4 == 0 && B > 0 && P > 0 && P % 4 == 0, "dppm_mlp_pair_fwd: H=%d B=%d P=%d", H, B, P);
  DPPM_REQUIRE((hid1 == nullptr) == (hid2 == nullptr), "dppm_mlp_pair_fwd: pass both hidden buffers or none");
  DPPM_REQUIRE(H == WS_H || hid1, "dppm_mlp_pair_fwd: H != 128 needs hid1/hid2 as scratch");
  DPPM_REQUIRE(aligned16(X) && aligned16(o1) && aligned16(o2) && x_bs % 4 == 0, "dppm_mlp_pair_fwd: planes must be 16-byte aligned");
  DPPM_REQUIRE(valid_prec(precision), "dppm_mlp_pair_fwd: precision=%d", precision);
  return mlp_pair(Geo{B, P, as_stream(stream)}, *q, H, X, x_bs, maskp, o1, o2, hid1, hid2, hid1 != nullptr, precision == DPPM_PREC_F32X6 ? DPPM_PREC_F32 : precision);
}

extern "C" size_t dppm_linear_wgrad_workspace_bytes(int M, int Nw, int B, int P) {
  if (M <= 0 || Nw <= 0 || B <= 0 || P <= 0) return 0;
  const size_t nz = (size_t)B * wgrad_nsplit(B, P);
  return (nz * M * Nw + nz * M + 64) * sizeof(float);
}

extern "C" int dppm_linear_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int M, int Nw, int B, int P, float* dW,
                                 int ldo, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  DPPM_REQUIRE(dY && X && dW && workspace, "dppm_linear_wgrad: null pointer");
  DPPM_REQUIRE(M > 0 && Nw > 0 && B > 0 && P > 0 && ldo >= Nw, "dppm_linear_wgrad: M=%d Nw=%d B=%d P=%d ldo=%d", M, Nw, B, P, ldo);
  DPPM_REQUIRE(workspace_bytes >= dppm_linear_wgrad_workspace_bytes(M, Nw, B, P), "dppm_linear_wgrad: workspace too small");
  const size_t nz = (size_t)B * wgrad_nsplit(B, P);
  float* slabs = static_cast<float*>(workspace);
  return wgrad_raw(Geo{B, P, as_stream(stream)}, M, dY, dy_bs, Nw, X, x_bs, dW, ldo, db, slabs, nz * M * Nw, slabs + nz * M * Nw);
}

extern "C" int dppm_inorm_fwd(const float* y, long y_bs, const float* maskp, float* u, long u_bs, int B, int C, int P,
                              float* mean_out, float* rstd_out, void* stream) {
  DPPM_REQUIRE(y && maskp && u, "dppm_inorm_fwd: null pointer");
  DPPM_REQUIRE(B > 0 && C > 0 && P > 0 && P % 4 == 0 && y_bs % 4 == 0 && u_bs % 4 == 0, "dppm_inorm_fwd: B=%d C=%d P=%d", B, C, P);
  DPPM_REQUIRE(aligned16(y) && aligned16(maskp) && aligned16(u), "dppm_inorm_fwd: buffers must be 16-byte aligned");
  launch_inorm_fwd(P, B * C, as_stream(stream), y, y_bs, maskp, u, u_bs, C, P, mean_out, rstd_out);
  return check_launch("dppm_inorm_fwd");
}

extern "C" int dppm_ppgn_forward(const dppm_ppgn_params* pr, const float* A, const float* mask, const float* sigma, int B,
                                 int N, float* logits, void* workspace, size_t workspace_bytes, int training, int precision,
                                 void* stream) {
  DPPM_TRY(check_params(pr, "dppm_ppgn_forward"));
  DPPM_REQUIRE(valid_prec(precision), "dppm_ppgn_forward: precision=%d", precision);
  const int prec = (training || precision == DPPM_PREC_F32X6) ? DPPM_PREC_F32 : precision;   // gradients are taken of the float32 forward
  DPPM_REQUIRE(A && mask && sigma && logits && workspace, "dppm_ppgn_forward: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_ppgn_forward: B=%d N=%d must be positive", B, N);
  DPPM_REQUIRE(((uintptr_t)workspace & 255) == 0, "dppm_ppgn_forward: workspace must be 256-byte aligned");
  const int H = pr->hidden, L = pr->num_layers;
  Ctx c{as_stream(stream), make_layout(B, N, H, L, training), static_cast<float*>(workspace)};
  const Layout& w = c.w;
  DPPM_REQUIRE(workspace_bytes >= w.total * sizeof(float), "dppm_ppgn_forward: workspace %zu < %zu bytes", workspace_bytes,
               w.total * sizeof(float));
  // small graphs (N <= 20, hidden 32: the reference's ego_small / community_small networks): one launch, one workgroup per graph,
  // activations in registers and LDS (dppm_smallgraph.hip).  Inference leaves the workspace untouched; training writes into it
  // exactly what dppm_ppgn_backward reads.
  if (sg_supported(pr, N)) {
    if (!training) return sg_forward(pr, A, mask, sigma, B, N, logits, nullptr, c.st, prec == DPPM_PREC_BF16X3 && sw(SW_DISABLE_MLP32S) == 0);
    SgSave sv{c.at(w.maskp), c.at(w.inv_norm), c.at(w.cat), c.at(w.u0), c.at(w.hid1), c.at(w.hid2), c.at(w.o1), c.at(w.o2), c.at(w.mm),
              c.at(w.rstd), c.at(w.R), c.at(w.act), w.S};
    return sg_forward(pr, A, mask, sigma, B, N, logits, &sv, c.st);
  }
  const int P = w.P;
  const long cat_bs = (long)w.CT * P, hs = (long)H * P;
  float* maskp = c.at(w.maskp);
  float* cat = c.at(w.cat);

  hipLaunchKernelGGL(prep_kernel, dim3(cdiv(P, 256) < 64 ? cdiv(P, 256) : 64, B), dim3(256), 0, c.st, A, mask, sigma, N, w.ld,
                     w.CT, cat, maskp, c.at(w.inv_norm));
  DPPM_TRY(check_launch("prep"));

  // Split modes, H = 128: the instance norm has no pass of its own.  m4 writes its raw result into the layer's CAT slot together
  // with per-tile statistics, norm_finalize_kernel turns them into a (scale, shift) table per graph and channel, and the three
  // consumers of the layer output (next layer's MLP pair and skip input, read-out head) apply it while staging.
  const int tpp = (P + WS_TP - 1) / WS_TP;
  const bool fuse_norm = use_wstat(H) && !training && prec != DPPM_PREC_F32 && !pr->layer_cond && sw(SW_DISABLE_NORMFUSE) == 0 &&
                         sw(SW_DISABLE_HEAD3) == 0 &&
                         (sw(SW_DISABLE_W4) == 0 ||     // the forms of dppm_wsplit.h keep per-graph tables in LDS
                          (w3_contig_ok((long)B * tpp, tpp, (int)(8 * w3_slots(2, (long)B * tpp))) &&
                           w3_contig_ok((long)B * tpp, tpp, (int)(8 * w3_slots(1, (long)B * tpp)))));
  // hidden 32, N > 20 (the wide kernels): the same folding in exact f32 -- lin32x2w_kernel leaves the raw result + group statistics,
  // mlp32w / lin32x2w / head32w apply the (scale, shift) tables to their sources
  const bool fuse32 = H == 32 && !training && !pr->layer_cond && sw(SW_DISABLE_NORMFUSE) == 0 && sw(SW_DISABLE_MLP32) == 0 && sw(SW_DISABLE_WIDE32) == 0 &&
                      P % 4 == 0 && L <= H32_MAXL;
  // in_lin (K = 2) folded into layer 0's consumers: u0 is never written (one plane-set less to write, two less to read per level)
  const bool fuse_inlin = fuse_norm && sw(SW_DISABLE_W4) == 0 && sw(SW_DISABLE_INLINFUSE) == 0;
  const InlSrc inl{pr->in_w, pr->in_b, cat, cat_bs};
  if (!fuse_inlin) {
    // in_lin: u0 = W_in (H x 2) * [A ; sigma I] + b     (model/ppgn.py:301)
    hipLaunchKernelGGL(inlin_kernel, dim3(cdiv(P, 1024), B), dim3(256), 0, c.st, cat, cat_bs, pr->in_w, pr->in_b, H, P, c.at(w.u0));
    DPPM_TRY(check_launch("in_lin"));
  }
  float* naff = (fuse_norm || fuse32) ? c.at(w.naff) : nullptr;
  const size_t aff_sz = (size_t)B * 2 * H;
  if (fuse_norm) {
    hipLaunchKernelGGL(aff_identity_kernel, dim3(cdiv((long)aff_sz, 256)), dim3(256), 0, c.st, naff + (size_t)L * aff_sz, (int)aff_sz);
    DPPM_TRY(check_launch("aff_identity"));
  }
  for (int l = 0; l < L; ++l) {
    const dppm_ppgn_layer& q = pr->layers_host[l];
    const float* x = l == 0 ? c.at(w.u0) : cat + (size_t)(2 + (l - 1) * H) * P;
    const long x_bs = l == 0 ? hs : cat_bs;
    float* hid1 = training ? c.at(w.hid1) + (size_t)l * w.S : c.at(w.tmp[0]);
    float* hid2 = training ? c.at(w.hid2) + (size_t)l * w.S : c.at(w.tmp[0]);
    float* o1 = training ? c.at(w.o1) + (size_t)l * w.S : c.at(w.tmp[1]);
    float* o2 = training ? c.at(w.o2) + (size_t)l * w.S : c.at(w.tmp[2]);
    float* mm = training ? c.at(w.mm) + (size_t)l * w.S : c.at(w.tmp[0]);
    float* y = c.at(w.tmp[3]);
    if (pr->layer_cond) {   // variant conditioning: the layer's scalar on the diagonal of the pad channel of its input
      float* xin = l == 0 ? c.at(w.u0) : cat + (size_t)(2 + (l - 1) * H) * P;
      hipLaunchKernelGGL(diag_set_kernel, dim3(B), dim3(256), 0, c.st, xin + (size_t)(H - 1) * P, x_bs, N, w.ld, pr->layer_cond + (size_t)l * B);
      DPPM_TRY(check_launch("diag_set"));
    }
    DPPM_TRY(mlp_pair(Geo{B, P, c.st}, q, H, x, x_bs, maskp, o1, o2, hid1, hid2, training != 0, prec,
                      (fuse_norm || fuse32) && l > 0 ? naff + (size_t)(l - 1) * aff_sz : nullptr, fuse_inlin && l == 0 ? &inl : nullptr));
    // per-channel contraction / sqrt(n)   (model/ppgn.py:62-64)
    DPPM_TRY(plane_mm(c.st, o1, o2, mm, B * H, N, w.ld, false, false, c.at(w.inv_norm), nullptr, H, training ? DPPM_PREC_F32X6 : prec));
    // m4 on cat(mm, x)   (model/ppgn.py:66-69)
    if (use_wstat(H)) {
      WsArgs wa{};
      wa.njobs = 1; wa.B = B; wa.P = P;
      WsJob& j = wa.job[0];
      j.X = mm; j.x_bs = hs; j.X2 = x; j.x2_bs = x_bs; j.W1 = q.m4_w; j.W2 = q.m4_w + H; j.ldw1 = 2 * H; j.ldw2 = 2 * H; j.b1 = q.m4_b;
      j.Y = y; j.y_bs = hs;
      if (fuse_norm) {   // raw result straight into the CAT slot, statistics beside it
        j.Y = cat + (size_t)(2 + l * H) * P; j.y_bs = cat_bs;
        wa.aff = naff + (size_t)(l > 0 ? l - 1 : L) * aff_sz;
        wa.stat = c.at(w.nstat); wa.cnt = c.at(w.ncnt); wa.maskstat = maskp;
        wa.dbg = sw(SW_DBG);
        if (fuse_inlin && l == 0) {   // the skip input u0 is formed while staging
          wa.inl_w = inl.w; wa.inl_b = inl.b; wa.inl_in = inl.in; wa.inl_bs = inl.bs;
          if (!w4_lin2_ok(wa)) return fail(DPPM_E_INVALID, "dppm_ppgn_forward: the in_lin form of m4 does not apply");
        }
      }
      // split inference modes: staging waves / multiplying waves (dppm_w4.hip); DPPM_DISABLE_W4=1 keeps the two-role form of dppm_wsplit.h
      if (!training && prec != DPPM_PREC_F32 && sw(SW_DISABLE_W4) == 0 && w4_lin2_ok(wa))
        DPPM_TRY(w4_lin2_run(wa, prec == DPPM_PREC_BF16X3 ? 3 : 1, c.st));
      else
      DPPM_TRY(ws_lin_p(wa, 2, false, prec, c.st));
      if (fuse_norm) {
        hipLaunchKernelGGL(norm_finalize_kernel, dim3(B * (WS_H / NF_CH)), dim3(NF_CH * NF_GRP), 0, c.st, (const float*)c.at(w.nstat), (const float*)c.at(w.ncnt), tpp,
                           naff + (size_t)l * aff_sz, WS_H);
        DPPM_TRY(check_launch("norm_finalize"));
        if (wa.dbg & 4) {
          launch_inorm_fwd(P, B * H, c.st, (const float*)(cat + (size_t)(2 + l * H) * P), cat_bs, (const float*)maskp, y, hs, H, P, c.at(w.nstat), c.at(w.nstat) + B * H);
          hipLaunchKernelGGL(aff_from_stats_kernel, dim3((B * H + 255) / 256), dim3(256), 0, c.st, (const float*)c.at(w.nstat), (const float*)(c.at(w.nstat) + B * H),
                             naff + (size_t)l * aff_sz, B * H);
        }
        continue;
      }
    } else if (H == 32 && wide32_ok(P, mm, hs, y, hs) && aligned16(x) && x_bs % 4 == 0) {
      if (fuse32 && !aligned16(maskp)) return fail(DPPM_E_INVALID, "dppm_ppgn_forward: mask planes must be 16-byte aligned");
      const bool prof = prof_enabled();
      if (prof) prof_before(c.st);
      float* yo = fuse32 ? cat + (size_t)(2 + l * H) * P : y;   // fuse32: the raw result straight into the CAT slot, statistics beside it
      const bool split32 = !training && prec == DPPM_PREC_BF16X3 && sw(SW_DISABLE_MLP32S) == 0;
      hipLaunchKernelGGL(split32 ? lin32x2w_kernel<true> : lin32x2w_kernel<false>, dim3(wide32_grid(B, P)), dim3(256), 0, c.st, (const float*)q.m4_w, 2 * H,
                         (const float*)q.m4_b, (const float*)mm, hs,
                         x, x_bs, yo, fuse32 ? cat_bs : hs, P, B, (const float*)(fuse32 && l > 0 ? naff + (size_t)(l - 1) * aff_sz : nullptr), (const float*)maskp,
                         fuse32 ? c.at(w.nstat) : (float*)nullptr, fuse32 ? c.at(w.ncnt) : (float*)nullptr);
      if (prof) prof_after(c.st, PROF_LINEAR, split32 ? AR_BF16X3 : AR_F32, 4.0 * 32 * 32 * (double)P * B, 3.0 * sizeof(float) * 32 * (double)P * B);
      DPPM_TRY(check_launch("lin32x2w"));
      if (fuse32) {
        hipLaunchKernelGGL(norm_finalize_kernel, dim3(B * (H / NF_CH)), dim3(NF_CH * NF_GRP), 0, c.st, (const float*)c.at(w.nstat), (const float*)c.at(w.ncnt), tpp,
                           naff + (size_t)l * aff_sz, H);
        DPPM_TRY(check_launch("norm_finalize"));
        continue;
      }
    } else {
      if (fuse32) return fail(DPPM_E_INVALID, "dppm_ppgn_forward: the folded instance norm of the hidden-32 path needs the wide kernels");
      LinArgs sx{q.m4_w + H, 2 * H, false, x, x_bs, H};
      DPPM_TRY(linear(c, H, LinArgs{q.m4_w, 2 * H, false, mm, hs, H}, &sx, q.m4_b, 0, nullptr, nullptr, 0, y, hs, false));
    }
    // masked instance norm, * mask; result lands in its slot of CAT   (model/ppgn.py:307,315-316)
    if (prof_enabled()) prof_before(c.st);
    launch_inorm_fwd(P, B * H, c.st, (const float*)y, hs, (const float*)maskp, cat + (size_t)(2 + l * H) * P, cat_bs, H, P, (float*)nullptr,
                     c.at(w.rstd) + (size_t)l * B * H);
    if (prof_enabled()) prof_after(c.st, PROF_NORM, AR_NONE, 2.0 * sizeof(float) * (double)w.S, 2.0 * sizeof(float) * (double)w.S);   // read y once, write u once
    DPPM_TRY(check_launch("inorm_fwd"));
  }
  // split modes, H = 128: layer_cat_lin, after_conv and final_lin in one pass over the layer outputs (dppm_head.h)
  const bool head_off = sw(SW_DISABLE_HEAD3) != 0;   // A/B switch for measurements and tests (read per call)
  if (use_wstat(H) && !training && prec != DPPM_PREC_F32 && !head_off) {
    HeadArgs ha{cat, cat_bs, reinterpret_cast<const uint4*>(c.at(w.himg)), pr->cat_w, w.CT, pr->cat_b, pr->after_b, pr->final_w, pr->final_b,
                logits, B, N, w.ld, P, L, fuse_norm ? naff : nullptr, fuse_norm ? maskp : nullptr};
    return head3_run(ha, pr->after_w, prec == DPPM_PREC_BF16X3 ? 3 : 1, c.st);
  }
  if (fuse32 && !(wide32_ok(P, cat, cat_bs, c.at(w.R), hs) && aligned16(c.at(w.act)))) return fail(DPPM_E_INVALID, "dppm_ppgn_forward: hidden-32 read-out: alignment");
  if (H == 32 && L <= H32_MAXL && wide32_ok(P, cat, cat_bs, c.at(w.R), hs) && aligned16(c.at(w.act))) {   // hidden-32 read-out in one pass
    const bool hsplit = !training && prec == DPPM_PREC_BF16X3 && sw(SW_DISABLE_MLP32S) == 0;
    const size_t lds = (size_t)(((2 + 32 * L) * 33 + 3) & ~3) * sizeof(float) + (hsplit ? (size_t)L * 4 * 64 * 16 : 0);
    auto hkern = hsplit ? head32w_kernel<true> : head32w_kernel<false>;
    DPPM_TRY(lds_limit(reinterpret_cast<const void*>(hkern), lds, "head32w"));
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    const int grid = wide32_grid(B, P) < 512 ? wide32_grid(B, P) : 512;   // every block stages the weights once
    hipLaunchKernelGGL(hkern, dim3(grid), dim3(256), lds, c.st, (const float*)cat, cat_bs, (const float*)pr->cat_w, w.CT, (const float*)pr->cat_b,
                       (const float*)pr->after_w, (const float*)pr->after_b, (const float*)pr->final_w, (const float*)pr->final_b,
                       training ? c.at(w.R) : (float*)nullptr, training ? c.at(w.act) : (float*)nullptr, hs, logits, N, w.ld, L, B,
                       (const float*)(fuse32 ? naff : nullptr), (const float*)maskp);
    if (prof) prof_after(c.st, PROF_LINEAR, hsplit ? AR_BF16X3 : AR_F32, 2.0 * 32 * ((double)w.CT + 32 + 1) * P * B,
                         sizeof(float) * ((double)w.CT * P + (training ? 64.0 * P : 0.0) + (double)N * N) * B);
    return check_launch("head32w");
  }
  // layer_cat_lin over cat([A, sigma I, u_1..u_L])   (model/ppgn.py:318,327)
  if (use_wstat(H)) {   // K = 2 + L*128: the layer outputs two at a time through the weight-stationary kernel, [A, sigma I] as a side input
    for (int l0 = 0; l0 < L; l0 += 2) {
      const int ns = L - l0 >= 2 ? 2 : 1;
      WsArgs wa{};
      wa.njobs = 1; wa.B = B; wa.P = P;
      WsJob& j = wa.job[0];
      j.X = cat + (size_t)(2 + l0 * H) * P; j.x_bs = cat_bs; j.W1 = pr->cat_w + 2 + (size_t)l0 * H; j.ldw1 = w.CT;
      if (ns == 2) { j.X2 = cat + (size_t)(2 + (l0 + 1) * H) * P; j.x2_bs = cat_bs; j.W2 = pr->cat_w + 2 + (size_t)(l0 + 1) * H; j.ldw2 = w.CT; }
      j.Y = c.at(w.R); j.y_bs = hs;
      if (l0 == 0) { j.b1 = pr->cat_b; wa.side = cat; wa.side_bs = cat_bs; wa.wside = pr->cat_w; wa.ldws = w.CT; }
      else j.accumulate = 1;
      DPPM_TRY(ws_lin_p(wa, ns, false, prec, c.st));
    }
  } else {
    DPPM_TRY(linear(c, H, LinArgs{pr->cat_w, w.CT, false, cat, cat_bs, w.CT}, nullptr, pr->cat_b, 0, nullptr, nullptr, 0, c.at(w.R), hs, false));
  }
  // out + LeakyReLU(after_conv(out)) ; final_lin   (model/ppgn.py:331-333)
  if (use_wstat(H)) {
    WsArgs wa{};
    wa.njobs = 1; wa.B = B; wa.P = P; wa.act = 1;
    WsJob& j = wa.job[0];
    j.X = c.at(w.R); j.x_bs = hs; j.W1 = pr->after_w; j.ldw1 = H; j.b1 = pr->after_b; j.Y = c.at(w.act); j.y_bs = hs;
    DPPM_TRY(ws_lin_p(wa, 1, false, training ? DPPM_PREC_F32X6 : prec, c.st));
  } else {
    DPPM_TRY(linear(c, H, LinArgs{pr->after_w, H, false, c.at(w.R), hs, H}, nullptr, pr->after_b, 1, nullptr, nullptr, 0, c.at(w.act), hs, false));
  }
  hipLaunchKernelGGL(final_fwd_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, c.st, c.at(w.R), c.at(w.act), pr->final_w, pr->final_b,
                     H, N, w.ld, logits);
  return check_launch("final_fwd");
}

// The whole sampling chain of small graphs in one launch (dppm_smallgraph.hip, CHAIN): see include/dppm_hip.h
extern "C" int dppm_ppgn_chain_supported(const dppm_ppgn_params* pr, int N) {
  return pr && pr->hidden > 0 && sg_supported(pr, N) && sw(SW_DISABLE_SGCHAIN) == 0 ? 1 : 0;
}
static int chain_entry(bool vlb, const char* name, const dppm_ppgn_params* pr, const float* flags, const float* mask, const float* level_table,
                       int T, uint64_t seed, uint32_t graph_id0, int B, int N, float* x_out, int precision, void* stream) {
  DPPM_TRY(check_params(pr, name));
  DPPM_REQUIRE(valid_prec(precision), "%s: precision=%d", name, precision);
  DPPM_REQUIRE(flags && mask && level_table && x_out, "%s: null pointer", name);
  DPPM_REQUIRE(B > 0 && N > 0 && T > 0, "%s: B=%d N=%d T=%d must be positive", name, B, N, T);
  DPPM_REQUIRE(sg_supported(pr, N), "%s: needs hidden 32, N <= %d, at most %d layers, no layer_cond (hidden=%d N=%d): "
               "run the chain level by level (dppm_ppgn_forward and the dppm_reverse_step_* entry points)", name, SG_MAX_NODES, SG_MAX_LAYERS, pr->hidden, N);
  const int prec = precision == DPPM_PREC_F32X6 ? DPPM_PREC_F32 : precision;
  return sg_chain(vlb, pr, flags, mask, level_table, T, seed, graph_id0, B, N, x_out, as_stream(stream),
                  prec == DPPM_PREC_BF16X3 && sw(SW_DISABLE_MLP32S) == 0);
}
extern "C" int dppm_ppgn_chain_simple(const dppm_ppgn_params* pr, const float* flags, const float* mask, const float* level_table,
                                      int T, uint64_t seed, uint32_t graph_id0, int B, int N, float* x_out, int precision,
                                      void* stream) {
  return chain_entry(false, "dppm_ppgn_chain_simple", pr, flags, mask, level_table, T, seed, graph_id0, B, N, x_out, precision, stream);
}
extern "C" int dppm_ppgn_chain_vlb(const dppm_ppgn_params* pr, const float* flags, const float* mask, const float* level_table,
                                   int T, uint64_t seed, uint32_t graph_id0, int B, int N, float* x_out, int precision,
                                   void* stream) {
  return chain_entry(true, "dppm_ppgn_chain_vlb", pr, flags, mask, level_table, T, seed, graph_id0, B, N, x_out, precision, stream);
}

extern "C" int dppm_ppgn_backward(const dppm_ppgn_params* pr, const dppm_ppgn_params* gr, const float* A, const float* mask,
                                  const float* sigma, int B, int N, const float* dlogits, float* dsigma, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  (void)A; (void)mask; (void)sigma;
  const int prec = DPPM_PREC_F32X6;   // float32-grade: exact-f32 MFMAs, or the three-piece split where a kernel has it
  DPPM_TRY(check_params(pr, "dppm_ppgn_backward(params)"));
  DPPM_TRY(check_params(gr, "dppm_ppgn_backward(grads)"));
  DPPM_REQUIRE(dlogits && workspace, "dppm_ppgn_backward: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_ppgn_backward: B=%d N=%d must be positive", B, N);
  DPPM_REQUIRE(gr->hidden == pr->hidden && gr->num_layers == pr->num_layers, "dppm_ppgn_backward: grads/params shape mismatch");
  DPPM_REQUIRE(!pr->layer_cond || gr->layer_cond, "dppm_ppgn_backward: params carry layer_cond, grads->layer_cond ([L][B]) must receive its gradient");
  const int H = pr->hidden, L = pr->num_layers;
  Ctx c{as_stream(stream), make_layout(B, N, H, L, 1), static_cast<float*>(workspace)};
  const Layout& w = c.w;
  DPPM_REQUIRE(workspace_bytes >= w.total * sizeof(float), "dppm_ppgn_backward: workspace %zu < %zu bytes", workspace_bytes,
               w.total * sizeof(float));
  const int P = w.P;
  const long cat_bs = (long)w.CT * P, hs = (long)H * P;
  float* maskp = c.at(w.maskp);
  float* cat = c.at(w.cat);
  float* dcat = c.at(w.dcat);
  float* dh = c.at(w.dh);
  float* dz = c.at(w.dz);
  float* dy = c.at(w.dy);
  float* dmm = c.at(w.dmm);
  float* do1 = c.at(w.do1);
  float* do2 = c.at(w.do2);
  float* du0 = c.at(w.du0);
  float* part = c.at(w.rsum) + (size_t)B * wgrad_nsplit(B, P) * H;   // [B*H] + [B]

  // head: final_lin and after_conv
  hipLaunchKernelGGL(final_bwd_kernel, dim3(B * H), dim3(256), 0, c.st, c.at(w.R), c.at(w.act), pr->final_w, dlogits, H, N, w.ld,
                     dh, dz, part, part + (size_t)B * H);
  DPPM_TRY(check_launch("final_bwd"));
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(H, RS_E) + 1), dim3(RS_E * RS_G), 0, c.st, (const float*)part, B, 1, H, gr->final_w, H,
                     (const float*)(part + (size_t)B * H), gr->final_b);
  DPPM_TRY(check_launch("reduce(dwfin, dbfin)"));
  const bool ws = use_wstat(H);
  // small networks: the head's weight gradients (after_conv, layer_cat_lin block by block) as 32 x 32 jobs of ONE launch behind dR,
  // in_lin's as one more at the end, every fold in the single launch after the layer loop (wgrad32_multi)
  W32Pending w32_pending{};
  const bool head32 = !ws && H == 32 && sw(SW_DISABLE_LIN32) == 0 && wgrad32_multi_ok(c) && L + 2 <= W32_MAXJOBS && aligned16(cat) && cat_bs % 4 == 0;
  if (!head32) DPPM_TRY(wgrad(c, H, dz, hs, H, c.at(w.R), hs, gr->after_w, H, gr->after_b));
  // dR = dh + W_ac^T dz   (accumulated into dh)
  if (ws) {
    WsArgs wa{};
    wa.njobs = 1; wa.B = B; wa.P = P;
    WsJob& j = wa.job[0];
    j.X = dz; j.x_bs = hs; j.W1 = pr->after_w; j.ldw1 = H; j.Y = dh; j.y_bs = hs; j.accumulate = 1;
    DPPM_TRY(ws_lin_p(wa, 1, true, prec, c.st));
  } else {
    DPPM_TRY(linear(c, H, LinArgs{pr->after_w, H, true, dz, hs, H}, nullptr, nullptr, 0, nullptr, nullptr, 0, dh, hs, true));
  }
  // layer_cat_lin
  if (ws) {   // one 128 x 128 block per layer output, then the two input columns
    for (int j = 0; j < L; ++j)
      DPPM_TRY(wgrad(c, H, dh, hs, H, cat + (size_t)(2 + j * H) * P, cat_bs, gr->cat_w + 2 + (size_t)j * H, w.CT, j == 0 ? gr->cat_b : nullptr));
    DPPM_TRY(wgrad(c, H, dh, hs, 2, cat, cat_bs, gr->cat_w, w.CT, nullptr));
  } else if (head32) {
    W32Job jobs[W32_MAXJOBS] = {{dz, hs, c.at(w.R), hs, gr->after_w, H, gr->after_b, 0}, {dh, hs, cat, cat_bs, gr->cat_w, w.CT, gr->cat_b, 2}};
    for (int j = 0; j < L; ++j) jobs[2 + j] = W32Job{dh, hs, cat + (size_t)(2 + j * H) * P, cat_bs, gr->cat_w + 2 + (size_t)j * H, w.CT, nullptr, 0};
    DPPM_TRY(wgrad32_multi(c, jobs, L + 2, &w32_pending));
  } else {
    DPPM_TRY(wgrad(c, H, dh, hs, w.CT, cat, cat_bs, gr->cat_w, w.CT, gr->cat_b));
  }
  if (ws) {   // d u_j = W_cat[:, 2 + jH ..]^T dh, one job per layer output (rows 0, 1 of dcat - the inputs - have no consumer)
    for (int j0 = 0; j0 < L; j0 += WS_MAX_JOBS) {
      WsArgs wa{};
      wa.njobs = L - j0 < WS_MAX_JOBS ? L - j0 : WS_MAX_JOBS; wa.B = B; wa.P = P;
      for (int k = 0; k < wa.njobs; ++k) {
        WsJob& j = wa.job[k];
        j.X = dh; j.x_bs = hs; j.W1 = pr->cat_w + 2 + (size_t)(j0 + k) * H; j.ldw1 = w.CT;
        j.Y = dcat + (size_t)(2 + (j0 + k) * H) * P; j.y_bs = cat_bs;
      }
      DPPM_TRY(ws_lin_p(wa, 1, true, prec, c.st));
    }
  } else if (head32) {   // rows 0, 1 of dcat (the inputs) have no consumer, as above
    const bool prof = prof_enabled();
    if (prof) prof_before(c.st);
    hipLaunchKernelGGL(lin32t_blocks_kernel, dim3(cdiv((long)B * cdiv(P, 16), 4)), dim3(256), 0, c.st, (const float*)(pr->cat_w + 2), w.CT, (const float*)dh, hs,
                       dcat + (size_t)2 * P, cat_bs, P, B, L);
    if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 2.0 * 32 * 32 * L * (double)P * B);
    DPPM_TRY(check_launch("lin32t_blocks(cat^T)"));
  } else {
    DPPM_TRY(linear(c, w.CT, LinArgs{pr->cat_w, w.CT, true, dh, hs, H}, nullptr, nullptr, 0, nullptr, nullptr, 0, dcat, cat_bs, false));
  }

  for (int l = L - 1; l >= 0; --l) {
    const dppm_ppgn_layer& q = pr->layers_host[l];
    const dppm_ppgn_layer& gq = gr->layers_host[l];
    const float* x = l == 0 ? c.at(w.u0) : cat + (size_t)(2 + (l - 1) * H) * P;
    const long x_bs = l == 0 ? hs : cat_bs;
    float* dx = l == 0 ? du0 : dcat + (size_t)(2 + (l - 1) * H) * P;
    const long dx_bs = l == 0 ? hs : cat_bs;
    const float* hid1 = c.at(w.hid1) + (size_t)l * w.S;
    const float* hid2 = c.at(w.hid2) + (size_t)l * w.S;
    const float* o1 = c.at(w.o1) + (size_t)l * w.S;
    const float* o2 = c.at(w.o2) + (size_t)l * w.S;
    const float* mm = c.at(w.mm) + (size_t)l * w.S;
    // instance norm
    launch_inorm_bwd(P, B * H, c.st, (const float*)(dcat + (size_t)(2 + l * H) * P), cat_bs, (const float*)(cat + (size_t)(2 + l * H) * P), cat_bs,
                     (const float*)maskp, (const float*)(c.at(w.rstd) + (size_t)l * B * H), dy, H, P);
    DPPM_TRY(check_launch("inorm_bwd"));
    // m4
    const bool multi = !ws && H == 32 && sw(SW_DISABLE_LIN32) == 0 && wgrad32_multi_ok(c);   // m4's gradients ride with the MLPs' below
    if (!multi) {
      DPPM_TRY(wgrad(c, H, dy, hs, H, mm, hs, gq.m4_w, 2 * H, gq.m4_b));
      DPPM_TRY(wgrad(c, H, dy, hs, H, x, x_bs, gq.m4_w + H, 2 * H, nullptr));
    }
    if (ws) {
      WsArgs wa{};
      wa.njobs = 2; wa.B = B; wa.P = P;
      WsJob& j0 = wa.job[0];
      j0.X = dy; j0.x_bs = hs; j0.W1 = q.m4_w; j0.ldw1 = 2 * H; j0.Y = dmm; j0.y_bs = hs;
      WsJob& j1 = wa.job[1];
      j1.X = dy; j1.x_bs = hs; j1.W1 = q.m4_w + H; j1.ldw1 = 2 * H; j1.Y = dx; j1.y_bs = dx_bs; j1.accumulate = l != 0;
      DPPM_TRY(ws_lin_p(wa, 1, true, prec, c.st));
    } else if (H == 32 && sw(SW_DISABLE_LIN32) == 0) {   // small networks: both halves of m4^T from one read of dy
      const bool prof = prof_enabled();
      if (prof) prof_before(c.st);
      hipLaunchKernelGGL(lin32t_kernel, dim3(cdiv((long)B * cdiv(P, 16), 4)), dim3(256), 0, c.st, (const float*)q.m4_w, 2 * H, (const float*)dy, hs,
                         (const float*)nullptr, 0L, dmm, hs, P, B, 0, (const float*)(q.m4_w + H), dx, dx_bs, l != 0 ? 1 : 0);
      if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 4.0 * 32 * 32 * (double)P * B);
      DPPM_TRY(check_launch("lin32t(m4^T)"));
    } else {
      DPPM_TRY(linear(c, H, LinArgs{q.m4_w, 2 * H, true, dy, hs, H}, nullptr, nullptr, 0, nullptr, nullptr, 0, dmm, hs, false));
      DPPM_TRY(linear(c, H, LinArgs{q.m4_w + H, 2 * H, true, dy, hs, H}, nullptr, nullptr, 0, nullptr, nullptr, 0, dx, dx_bs, l != 0));
    }
    // contraction: do1 = dmm o2^T / sqrt(n) * mask ; do2 = o1^T dmm / sqrt(n) * mask
    if (!ws && H == 32 && plane_mm_pair_ok(N) && sw(SW_DISABLE_LIN32) == 0 && aligned16(dmm) && aligned16(o1) && aligned16(o2) && aligned16(do1) &&
        aligned16(do2)) {   // small networks: both products in one launch (exact-f32 MFMAs, the arithmetic the two calls below use at this size)
      const PlaneArgs p1{dmm, o2, do1, N, w.ld, c.at(w.inv_norm), maskp, H}, p2{o1, dmm, do2, N, w.ld, c.at(w.inv_norm), maskp, H};
      DPPM_TRY(plane_mm_pair_launch(p1, p2, B * H, c.st));
    } else {
    DPPM_TRY(plane_mm(c.st, dmm, o2, do1, B * H, N, w.ld, false, true, c.at(w.inv_norm), maskp, H, DPPM_PREC_F32X6));
    DPPM_TRY(plane_mm(c.st, o1, dmm, do2, B * H, N, w.ld, true, false, c.at(w.inv_norm), maskp, H, DPPM_PREC_F32X6));
    }
    // m1 (dy is free again: reuse it for the pre-activation gradient)
    float* dpre = dy;
    const float* dout[2] = {do1, do2}; const float* hid[2] = {hid1, hid2};
    const float* wl0[2] = {q.m1_w0, q.m2_w0}; const float* wl1[2] = {q.m1_w1, q.m2_w1};
    float* gw0[2] = {gq.m1_w0, gq.m2_w0}; float* gb0[2] = {gq.m1_b0, gq.m2_b0};
    float* gw1[2] = {gq.m1_w1, gq.m2_w1}; float* gb1[2] = {gq.m1_b1, gq.m2_b1};
    const bool m32_off = sw(SW_DISABLE_LIN32) != 0;
    if (!ws && H == 32 && !m32_off) {   // small networks: both MLP backward chains in one launch (dpre_1 -> dy, dpre_2 -> dmm, both free here)
      // multi: dpre_1 goes to dz instead (the head's after_conv gradient: scratch of this call, free since the head's backward), so
      // that dy is still m4's output gradient afterwards and ALL six weight gradients of the layer ride one launch behind the MLP chains
      float* dpre1 = multi ? dz : dy;
      if (!multi) {
        DPPM_TRY(wgrad(c, H, dout[0], hs, H, hid[0], hs, gw1[0], H, gb1[0]));
        DPPM_TRY(wgrad(c, H, dout[1], hs, H, hid[1], hs, gw1[1], H, gb1[1]));
      }
      const bool prof = prof_enabled();
      if (prof) prof_before(c.st);
      hipLaunchKernelGGL(mlp32_bwd_kernel, dim3(cdiv((long)B * cdiv(P, 16), 4)), dim3(256), 0, c.st, do1, do2, hid1, hid2, hs, wl1[0], wl1[1], wl0[0],
                         wl0[1], dpre1, dmm, dx, dx_bs, P, B);
      if (prof) prof_after(c.st, PROF_LINEAR, AR_F32, 8.0 * 32 * 32 * (double)P * B);
      DPPM_TRY(check_launch("mlp32_bwd"));
      if (multi) {
        const W32Job jobs[6] = {{dy, hs, mm, hs, gq.m4_w, 2 * H, gq.m4_b}, {dy, hs, x, x_bs, gq.m4_w + H, 2 * H, nullptr},
                                {dout[0], hs, hid[0], hs, gw1[0], H, gb1[0]}, {dout[1], hs, hid[1], hs, gw1[1], H, gb1[1]},
                                {dpre1, hs, x, x_bs, gw0[0], H, gb0[0]}, {dmm, hs, x, x_bs, gw0[1], H, gb0[1]}};
        DPPM_TRY(wgrad32_multi(c, jobs, 6, &w32_pending));
      } else {
        DPPM_TRY(wgrad(c, H, dy, hs, H, x, x_bs, gw0[0], H, gb0[0]));
        DPPM_TRY(wgrad(c, H, dmm, hs, H, x, x_bs, gw0[1], H, gb0[1]));
      }
    } else
    for (int k = 0; k < 2; ++k) {
      DPPM_TRY(wgrad(c, H, dout[k], hs, H, hid[k], hs, gw1[k], H, gb1[k]));
      if (ws) {   // dpre = (W1^T dout) * act'(hid) ; dx += W0^T dpre, the intermediate never leaves the CU except as the saved copy
        WsArgs wa{};
        wa.njobs = 1; wa.B = B; wa.P = P;
        WsJob& j = wa.job[0];
        j.X = dout[k]; j.x_bs = hs; j.W1 = wl1[k]; j.ldw1 = H; j.W2 = wl0[k]; j.ldw2 = H; j.Hsaved = hid[k]; j.hs_bs = hs;
        j.Mid = dpre; j.mid_bs = hs; j.Y = dx; j.y_bs = dx_bs; j.accumulate = 1;
        // two launches of the three-piece-split Linear (dppm_w6.hip): dpre has to be written anyway (the weight gradients read it)
        WsArgs l1{}, l2{};
        l1.njobs = l2.njobs = 1; l1.B = l2.B = B; l1.P = l2.P = P;
        WsJob& a1 = l1.job[0]; WsJob& a2 = l2.job[0];
        a1.X = dout[k]; a1.x_bs = hs; a1.W1 = wl1[k]; a1.ldw1 = H; a1.Hsaved = hid[k]; a1.hs_bs = hs; a1.Y = dpre; a1.y_bs = hs;
        a2.X = dpre; a2.x_bs = hs; a2.W1 = wl0[k]; a2.ldw1 = H; a2.Y = dx; a2.y_bs = dx_bs; a2.accumulate = 1;
        if (sw(SW_DISABLE_W6) == 0 && sw(SW_DISABLE_W6MLP) == 0 && w6_lin_ok(l1, 1) && w6_lin_ok(l2, 1)) {
          DPPM_TRY(w6_lin_run(l1, 1, true, c.st, PROF_WS_MLP2));
          DPPM_TRY(w6_lin_run(l2, 1, true, c.st, PROF_WS_MLP2));
        } else
        DPPM_TRY(ws_mlp_backward(wa, c.st));
      } else {
        DPPM_TRY(linear(c, H, LinArgs{wl1[k], H, true, dout[k], hs, H}, nullptr, nullptr, 0, nullptr, hid[k], hs, dpre, hs, false));
      }
      DPPM_TRY(wgrad(c, H, dpre, hs, H, x, x_bs, gw0[k], H, gb0[k]));
      if (!ws) DPPM_TRY(linear(c, H, LinArgs{wl0[k], H, true, dpre, hs, H}, nullptr, nullptr, 0, nullptr, nullptr, 0, dx, dx_bs, true));
    }
    if (pr->layer_cond) {   // dx is complete here: readout, m4 skip and both MLPs have been added
      hipLaunchKernelGGL(diag_sum_kernel, dim3(B), dim3(64), 0, c.st, (const float*)(dx + (size_t)(H - 1) * P), dx_bs, N, w.ld, const_cast<float*>(gr->layer_cond) + (size_t)l * B);
      DPPM_TRY(check_launch("diag_sum"));
    }
  }
  // in_lin
  if (head32) {
    const W32Job job{du0, hs, cat, cat_bs, gr->in_w, 2, gr->in_b, 2};   // (the first 32 planes of CAT; only the two input columns are stored)
    DPPM_TRY(wgrad32_multi(c, &job, 1, &w32_pending));
  }
  DPPM_TRY(wgrad32_flush(c, w32_pending));
  if (!head32) DPPM_TRY(wgrad(c, H, du0, hs, 2, cat, cat_bs, gr->in_w, 2, gr->in_b));
  if (dsigma) {   // dh still holds dR (nothing after the head reuses it)
    hipLaunchKernelGGL(dsigma_kernel, dim3(B), dim3(256), 0, c.st, (const float*)du0, (const float*)dh, pr->in_w, pr->cat_w, w.CT, H, N, w.ld, dsigma);
    DPPM_TRY(check_launch("dsigma"));
  }
  return DPPM_OK;
}
