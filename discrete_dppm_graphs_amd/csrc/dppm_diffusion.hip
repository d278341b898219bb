// Discrete edge-flip diffusion kernels for gfx950: forward noising q(A_t|A_0), x_T init, the two reverse steps
// and the two losses.  Every kernel is HBM-bound elementwise work on [B][N][N] float32 tensors; the symmetric
// structure (one Bernoulli draw per unordered pair, mirrored) is handled per 32x32 tile PAIR: a workgroup
// computes the strict-upper values of tile (ti,tj), keeps them in LDS and writes both the tile and its mirror
// image (tj,ti) with coalesced rows, so no transposed global access is ever issued.
//
// This translation unit is compiled with -ffp-contract=off: the reverse-step arithmetic must reproduce the
// reference's float32 operation order bit for bit (no FMA contraction).  File:line citations are to the
// reference tree (kilianhae/discrete_DPPM_Graphs).
#include <mutex>
#include "dppm_common.h"
#include "dppm_philox.h"

namespace dppm {
namespace {

constexpr int TILE = 32;
constexpr int TROWS = 8;   // 32 x 8 threads, each thread walks 4 rows of the tile

struct RngArgs {
  const float* U;      // explicit uniforms [B][N][N] or nullptr
  uint64_t seed;
  uint32_t graph_id0;
  uint32_t offset;
};

__device__ __forceinline__ float draw(const RngArgs& rng, int b, int i, int j, int N) {
  if (rng.U) return rng.U[((size_t)b * N + i) * N + j];
  return philox_uniform(rng.seed, rng.graph_id0 + (uint32_t)b, rng.offset, (uint32_t)i, (uint32_t)j);
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------
// tile-pair driver.  Op provides:
//   void  stage(b, ti, tj, ...)              optional loads of mirror tiles into LDS (via lower())
//   float upper(b, i, j, ii, jj)             value of the unordered pair (i<j)
//   void  write(b, i, j, v)                  store for output position (i,j) given the pair value v
// ---------------------------------------------------------------------------------------------------------
template <class Op>
__global__ __launch_bounds__(TILE* TROWS) void tilepair_kernel(Op op, int N, int ntile) {
  const int ti = blockIdx.x / ntile, tj = blockIdx.x % ntile;
  if (ti > tj) return;   // whole workgroup leaves together
  const int b = blockIdx.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  __shared__ float sv[TILE][TILE + 1];
  __shared__ float s0[TILE][TILE + 1];   // mirror tile of operand 0 (logits), indexed [row in tj-tile][col in ti-tile]
  __shared__ float s1[TILE][TILE + 1];   // mirror tile of operand 1 (A_t)

  if (Op::kMirrorOperands > 0) {
#pragma unroll
    for (int r = ty; r < TILE; r += TROWS) {
      const int j = tj * TILE + r, i = ti * TILE + tx;   // element (j,i) of the mirror tile, coalesced along i
      const bool ok = (j < N) && (i < N);
      s0[r][tx] = ok ? op.mirror0(b, j, i) : 0.f;
      if (Op::kMirrorOperands > 1) s1[r][tx] = ok ? op.mirror1(b, j, i) : 0.f;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = ty; r < TILE; r += TROWS) {
    const int i = ti * TILE + r, j = tj * TILE + tx;
    float v = 0.f;
    if (i < N && j < N && i < j)
      v = op.upper(b, i, j, Op::kMirrorOperands > 0 ? s0[tx][r] : 0.f, Op::kMirrorOperands > 1 ? s1[tx][r] : 0.f);
    sv[r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < TILE; r += TROWS) {
    {   // tile (ti,tj)
      const int i = ti * TILE + r, j = tj * TILE + tx;
      if (i < N && j < N) {
        float v = 0.f;
        if (i < j) v = sv[r][tx];
        else if (i > j && ti == tj) v = sv[tx][r];
        op.write(b, i, j, v);
      }
    }
    if (ti != tj) {   // mirror tile (tj,ti): rows of the tj range, columns of the ti range
      const int j = tj * TILE + r, i = ti * TILE + tx;
      if (i < N && j < N) op.write(b, j, i, sv[tx][r]);
    }
  }
}

template <class Op>
int launch_tilepair(const Op& op, int B, int N, hipStream_t st, const char* what) {
  const int ntile = cdiv(N, TILE);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(tilepair_kernel<Op>), dim3(ntile * ntile, B), dim3(TILE, TROWS), 0, st, op, N, ntile);
  return check_launch(what);
}

// ---------------------------------------------------------------------------------------------------------
// q(A_t | A_0)      utils/graph_utils.py:155-165 / :236-246 ; masks :7-19
// ---------------------------------------------------------------------------------------------------------
struct FlipOp {
  static constexpr int kMirrorOperands = 0;
  const float* A0; const float* flags; const double* sigma_dev; double sigma_scalar;
  RngArgs rng; int N; float* A_t; float* target;
  __device__ float mirror0(int, int, int) const { return 0.f; }
  __device__ float mirror1(int, int, int) const { return 0.f; }
  __device__ float upper(int b, int i, int j, float, float) const {
    const double s = sigma_dev ? sigma_dev[b] : sigma_scalar;
    const float p_keep = (float)(1.0 - s), p_flip = (float)s;   // rounded once from double, like full_like(x, 1-sigma)
    const float a0 = A0[((size_t)b * N + i) * N + j];
    const float p = a0 > 0.5f ? p_keep : p_flip;
    return draw(rng, b, i, j, N) < p ? 1.f : 0.f;
  }
  __device__ void write(int b, int i, int j, float v) const {
    const size_t idx = ((size_t)b * N + i) * N + j;
    const float fi = flags[b * N + i], fj = flags[b * N + j];
    if (target) target[idx] = fabsf(-A0[idx] + v) * fi * fj;   // read A0 before A_t may overwrite it (aliasing allowed)
    A_t[idx] = v * fi * fj;
  }
};

// ---------------------------------------------------------------------------------------------------------
// x_T      sample_ppgn_simple.py:31-41
// ---------------------------------------------------------------------------------------------------------
struct InitOp {
  static constexpr int kMirrorOperands = 0;
  const float* flags; RngArgs rng; int N; float* out;
  __device__ float mirror0(int, int, int) const { return 0.f; }
  __device__ float mirror1(int, int, int) const { return 0.f; }
  __device__ float upper(int b, int i, int j, float, float) const {
    const float p = (flags[b * N + i] > 0.5f && flags[b * N + j] > 0.5f) ? 0.5f : 0.f;
    return draw(rng, b, i, j, N) < p ? 1.f : 0.f;
  }
  __device__ void write(int b, int i, int j, float v) const { out[((size_t)b * N + i) * N + j] = v; }
};

// ---------------------------------------------------------------------------------------------------------
// simple reverse step      sample_ppgn_simple.py:58-68
// ---------------------------------------------------------------------------------------------------------
struct RevSimpleOp {
  static constexpr int kMirrorOperands = 1;
  const float* A_t; const float* logits; int is_probs; const float* flags; RngArgs rng; int N; float* out;
  __device__ float prob(int b, int i, int j) const {
    const float x = logits[((size_t)b * N + i) * N + j];
    return is_probs ? x : sigmoidf_dev(x);
  }
  __device__ float mirror0(int b, int j, int i) const { return prob(b, j, i); }
  __device__ float mirror1(int, int, int) const { return 0.f; }
  __device__ float upper(int b, int i, int j, float r_ji, float) const {
    float r = (prob(b, i, j) + r_ji) / 2.f;
    const float m = (flags[b * N + i] >= 0.9f && flags[b * N + j] >= 0.9f) ? 1.f : 0.f;   // generate_mask
    r = r * m;
    return draw(rng, b, i, j, N) < r ? 1.f : 0.f;
  }
  __device__ void write(int b, int i, int j, float v) const {
    const size_t idx = ((size_t)b * N + i) * N + j;
    const float a = A_t[idx];
    const float inter = v > 0.5f ? a - 1.f : a;
    out[idx] = inter < -0.5f ? inter + 2.f : inter;
  }
};

// ---------------------------------------------------------------------------------------------------------
// VLB reverse step      sample_ppgn_vlb.py:82-107
// ---------------------------------------------------------------------------------------------------------
struct VlbConsts { float one_m_b, b, st1, one_m_st1, st, one_m_st; };

__device__ __forceinline__ VlbConsts make_vlb_consts(double st, double beta, double st1) {
  VlbConsts c;
  c.one_m_b = (float)(1.0 - beta); c.b = (float)beta;
  c.st1 = (float)st1; c.one_m_st1 = (float)(1.0 - st1);
  c.st = (float)st; c.one_m_st = (float)(1.0 - st);
  return c;
}

// p(A_{t-1} = 1 | A_t) for one entry, in the reference's operation order (sample_ppgn_vlb.py:89-101).
__device__ __forceinline__ float posterior_mix(float a_t, float r_sym, const VlbConsts& c) {
  const bool on = a_t > 0.5f;
  const float s = on ? 1.f - r_sym : r_sym;
  const float mult1 = on ? c.one_m_b : c.b;
  const float div0 = on ? c.st : c.one_m_st;
  float p = (1.f - s) * mult1 * c.st1 / div0;
  const float div1 = on ? c.one_m_st : c.st;
  p = p + s * mult1 * c.one_m_st1 / div1;
  return p;
}

struct RevVlbOp {
  static constexpr int kMirrorOperands = 2;
  const float* A_t; const float* logits; int is_probs; const float* flags;
  double st, beta, st1; RngArgs rng; int N; float* out;
  __device__ float prob(int b, int i, int j) const {
    const float x = logits[((size_t)b * N + i) * N + j];
    return is_probs ? x : sigmoidf_dev(x);
  }
  __device__ float mirror0(int b, int j, int i) const { return prob(b, j, i); }
  __device__ float mirror1(int b, int j, int i) const { return A_t[((size_t)b * N + j) * N + i]; }
  __device__ float upper(int b, int i, int j, float r_ji, float a_ji) const {
    const VlbConsts c = make_vlb_consts(st, beta, st1);
    const float r_ij = prob(b, i, j);
    const float r_up = (r_ij + r_ji) / 2.f;
    const float r_lo = (r_ji + r_ij) / 2.f;
    const float p_ij = posterior_mix(A_t[((size_t)b * N + i) * N + j], r_up, c);
    const float p_ji = posterior_mix(a_ji, r_lo, c);
    float q = (p_ij + p_ji) / 2.f;
    const float m = (flags[b * N + i] >= 0.9f && flags[b * N + j] >= 0.9f) ? 1.f : 0.f;
    q = q * m;
    return draw(rng, b, i, j, N) < q ? 1.f : 0.f;
  }
  __device__ void write(int b, int i, int j, float v) const { out[((size_t)b * N + i) * N + j] = v; }
};

// ---------------------------------------------------------------------------------------------------------
// mask      utils/graph_utils.py:312-319
// ---------------------------------------------------------------------------------------------------------
__global__ void mask_kernel(const float* __restrict__ flags, int N, float* __restrict__ mask, size_t total) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx % N);
    const size_t bi = idx / N;   // b*N + i
    const size_t b = bi / N;
    mask[idx] = (flags[bi] >= 0.9f && flags[b * N + j] >= 0.9f) ? 1.f : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// losses.  One pass: per-element loss + gradient, wavefront/LDS reduction to one double per workgroup, then a
// single-workgroup second pass folds the partials in a fixed order (bitwise reproducible).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_sum(double v) {
  __shared__ double part[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) part[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)((blockDim.x + 63) >> 6); ++k) s += part[k];
  return s;   // valid in thread 0
}

// BCE-with-logits of one entry, numerically stable form: max(x,0) - x*y + log1p(exp(-|x|))
__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))); }

// ppgn_simple.py:25-34
__global__ __launch_bounds__(256) void loss_bce_kernel(const float* __restrict__ score, const float* __restrict__ target,
                                                       const double* __restrict__ sigma, const float* __restrict__ mask,
                                                       int N, int B_global, float* __restrict__ dscore,
                                                       double* __restrict__ partials) {
  const int b = blockIdx.y;
  const size_t NN = (size_t)N * N;
  const double w = 1.0 - 2.0 * sigma[b] + 1.0 / (double)B_global;
  const double inv_cnt = 1.0 / ((double)B_global * (double)NN);
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / N), j = (int)(e % N);
    const size_t idx = b * NN + e;
    // sum_ij sym(l)_ij m_ij == sum_ij l_ij (m_ij + m_ji)/2 for any mask
    const double msym = 0.5 * ((double)mask[idx] + (double)mask[b * NN + (size_t)j * N + i]);
    const float x = score[idx], y = target[idx];
    acc += (double)bce_logits(x, y) * w * msym;
    if (dscore) dscore[idx] = (float)((double)(sigmoidf_dev(x) - y) * w * msym * inv_cnt);
  }
  const double s = block_reduce_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = s * inv_cnt;
}

// ppgn_vlb.py:79-125
__global__ __launch_bounds__(256) void loss_vlb_kernel(const float* __restrict__ score, const float* __restrict__ A_t,
                                                       const float* __restrict__ A_0, const double* __restrict__ sched,
                                                       const float* __restrict__ mask, int N, float* __restrict__ dscore,
                                                       double* __restrict__ partials) {
  const int b = blockIdx.y;
  const size_t NN = (size_t)N * N;
  const VlbConsts c = make_vlb_consts(sched[3 * b + 0], sched[3 * b + 1], sched[3 * b + 2]);
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / N), j = (int)(e % N);
    const size_t idx = b * NN + e;
    const float msym = (i == j) ? 0.f : 0.5f * (mask[idx] + mask[b * NN + (size_t)j * N + i]);
    const float at = A_t[idx], a0 = A_0[idx];
    const bool on = at > 0.5f, on0 = a0 > 0.5f;
    // q(A_{t-1}=1 | A_t, A_0)   (:89-93)
    const float mult1 = on ? c.one_m_b : c.b;
    const float mult2 = on0 ? c.one_m_st1 : c.st1;
    const float div = (on != on0) ? c.st : c.one_m_st;
    const float q = mult1 * mult2 / div;
    // p(A_{t-1}=1 | A_t) from the flip logit   (:95-109)
    const float sg = sigmoidf_dev(score[idx]);
    const float p = posterior_mix(at, sg, c);
    // 2-class KL with torch's xlogy convention for a zero target   (:114-119)
    float l = 0.f;
    if (q > 0.f) l += q * (logf(q) - logf(p));
    const float q1 = 1.f - q, p1 = 1.f - p;
    if (q1 > 0.f) l += q1 * (logf(q1) - logf(p1));
    acc += (double)(l * msym);
    if (dscore) {
      // dl/dp = -q/p + (1-q)/(1-p);  dp/ds = c1 - c0;  ds/dz = +-sg(1-sg)
      const float div0 = on ? c.st : c.one_m_st, div1 = on ? c.one_m_st : c.st;
      const float c0 = mult1 * c.st1 / div0, c1 = mult1 * c.one_m_st1 / div1;
      float dldp = 0.f;
      if (q > 0.f) dldp -= q / p;
      if (q1 > 0.f) dldp += q1 / p1;
      const float dsdz = (on ? -1.f : 1.f) * sg * (1.f - sg);
      dscore[idx] = dldp * (c1 - c0) * dsdz * msym;
    }
  }
  const double s = block_reduce_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

template <typename OutT>
__global__ __launch_bounds__(256) void fold_partials_kernel(const double* __restrict__ partials, int n, OutT* __restrict__ out) {
  double acc = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) acc += partials[k];
  const double s = block_reduce_sum(acc);
  if (threadIdx.x == 0) *out = (OutT)s;
}

constexpr int kLossBlocksPerGraph = 8;

}  // namespace
}  // namespace dppm

// Scratch of dppm_loss_bce_fwd_bwd_noalloc (include/dppm_hip.h): the per-block partial sums the workspace argument carries in
// dppm_loss_bce_fwd_bwd, for up to DPPM_NOALLOC_MAX_GRAPHS graphs per call (256 KiB).  Each GPU the library runs on holds its own
// copy of this array at its own address until the process ends.  Host code never uses the NAME below as an address (that is a
// host-side stand-in): the device address is looked up once per device with hipGetSymbolAddress and kept in noalloc_scratch_ptr.
static __device__ __align__(256) double loss_noalloc_scratch[DPPM_NOALLOC_MAX_GRAPHS * 8];

namespace dppm {
namespace {
constexpr int kMaxDevices = 64;
double* g_noalloc_ptr[kMaxDevices];      // per device: address of that device's copy of loss_noalloc_scratch (null = not looked up yet)
std::mutex g_noalloc_mutex;

// address of the current device's scratch array, looked up on the first call on that device (with that device current)
int noalloc_scratch_ptr(double** out) {
  int devid = -1;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= kMaxDevices)
    return fail(DPPM_E_LAUNCH, "dppm_loss_bce_fwd_bwd_noalloc: cannot identify the current device (%d)", devid);
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_ptr[devid]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(loss_noalloc_scratch));
    if (e != hipSuccess || !p)
      return fail(DPPM_E_LAUNCH, "dppm_loss_bce_fwd_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", devid, hipGetErrorString(e));
    g_noalloc_ptr[devid] = static_cast<double*>(p);
  }
  *out = g_noalloc_ptr[devid];
  return DPPM_OK;
}
}  // namespace
}  // namespace dppm

using namespace dppm;

extern "C" int dppm_generate_mask(const float* flags, int B, int N, float* mask, void* stream) {
  DPPM_REQUIRE(flags && mask, "dppm_generate_mask: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_generate_mask: B=%d N=%d must be positive", B, N);
  const size_t total = (size_t)B * N * N;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(mask_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), flags, N, mask, total);
  return check_launch("dppm_generate_mask");
}

extern "C" int dppm_edgeflip_fwd(const float* A0, const float* flags, const double* sigma_dev, double sigma_scalar,
                                 const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset, int B, int N,
                                 float* A_t, float* target, void* stream) {
  DPPM_REQUIRE(A0 && flags && A_t, "dppm_edgeflip_fwd: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_edgeflip_fwd: B=%d N=%d must be positive", B, N);
  DPPM_REQUIRE(sigma_dev || (sigma_scalar >= 0.0 && sigma_scalar <= 1.0), "dppm_edgeflip_fwd: sigma %g outside [0,1]", sigma_scalar);
  DPPM_REQUIRE(target != A0 && (target == nullptr || target != A_t), "dppm_edgeflip_fwd: target must not alias A0/A_t");
  FlipOp op{A0, flags, sigma_dev, sigma_scalar, RngArgs{U, seed, graph_id0, offset}, N, A_t, target};
  return launch_tilepair(op, B, N, as_stream(stream), "dppm_edgeflip_fwd");
}

extern "C" int dppm_init_xT(const float* flags, const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset,
                            int B, int N, float* A_T, void* stream) {
  DPPM_REQUIRE(flags && A_T, "dppm_init_xT: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_init_xT: B=%d N=%d must be positive", B, N);
  InitOp op{flags, RngArgs{U, seed, graph_id0, offset}, N, A_T};
  return launch_tilepair(op, B, N, as_stream(stream), "dppm_init_xT");
}

extern "C" int dppm_reverse_step_simple(const float* A_t, const float* logits, int is_probs, const float* flags,
                                        const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset, int B, int N,
                                        float* out, void* stream) {
  DPPM_REQUIRE(A_t && logits && flags && out, "dppm_reverse_step_simple: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_reverse_step_simple: B=%d N=%d must be positive", B, N);
  RevSimpleOp op{A_t, logits, is_probs, flags, RngArgs{U, seed, graph_id0, offset}, N, out};
  return launch_tilepair(op, B, N, as_stream(stream), "dppm_reverse_step_simple");
}

extern "C" int dppm_reverse_step_vlb(const float* A_t, const float* logits, int is_probs, const float* flags,
                                     double sigmatilde_t, double beta_t, double sigmatilde_t1, const float* U,
                                     uint64_t seed, uint32_t graph_id0, uint32_t offset, int B, int N, float* out,
                                     void* stream) {
  DPPM_REQUIRE(A_t && logits && flags && out, "dppm_reverse_step_vlb: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_reverse_step_vlb: B=%d N=%d must be positive", B, N);
  DPPM_REQUIRE(out != A_t, "dppm_reverse_step_vlb: out must not alias A_t (mirror entries are re-read)");
  RevVlbOp op{A_t, logits, is_probs, flags, sigmatilde_t, beta_t, sigmatilde_t1, RngArgs{U, seed, graph_id0, offset}, N, out};
  return launch_tilepair(op, B, N, as_stream(stream), "dppm_reverse_step_vlb");
}

extern "C" size_t dppm_loss_workspace_bytes(int B, int N) {
  (void)N;
  return B > 0 ? (size_t)B * kLossBlocksPerGraph * sizeof(double) : 0;
}

extern "C" int dppm_loss_bce_fwd_bwd(const float* score, const float* target, const double* sigma_dev, const float* mask,
                                     int B, int N, int B_global, double* loss_out, float* dscore, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DPPM_REQUIRE(score && target && sigma_dev && mask && loss_out && workspace, "dppm_loss_bce_fwd_bwd: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0 && B_global >= B, "dppm_loss_bce_fwd_bwd: B=%d N=%d B_global=%d", B, N, B_global);
  DPPM_REQUIRE(workspace_bytes >= dppm_loss_workspace_bytes(B, N), "dppm_loss_bce_fwd_bwd: workspace %zu < %zu bytes",
               workspace_bytes, dppm_loss_workspace_bytes(B, N));
  double* partials = static_cast<double*>(workspace);
  hipLaunchKernelGGL(loss_bce_kernel, dim3(kLossBlocksPerGraph, B), dim3(256), 0, as_stream(stream), score, target,
                     sigma_dev, mask, N, B_global, dscore, partials);
  DPPM_TRY(check_launch("dppm_loss_bce_fwd_bwd"));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fold_partials_kernel<double>), dim3(1), dim3(256), 0, as_stream(stream), partials,
                     B * kLossBlocksPerGraph, loss_out);
  return check_launch("dppm_loss_bce_fwd_bwd(fold)");
}

extern "C" int dppm_loss_bce_fwd_bwd_noalloc(const float* score, const float* target, const double* sigma_dev, const float* mask,
                                             int B, int N, int B_global, double* loss_out, float* dscore, void* stream) {
  DPPM_REQUIRE(score && target && sigma_dev && mask && loss_out, "dppm_loss_bce_fwd_bwd_noalloc: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0 && B_global >= B, "dppm_loss_bce_fwd_bwd_noalloc: B=%d N=%d B_global=%d", B, N, B_global);
  DPPM_REQUIRE(B <= DPPM_NOALLOC_MAX_GRAPHS, "dppm_loss_bce_fwd_bwd_noalloc: B=%d exceeds the built-in scratch (at most %d graphs per call); "
               "use dppm_loss_bce_fwd_bwd with a workspace", B, DPPM_NOALLOC_MAX_GRAPHS);
  static_assert(kLossBlocksPerGraph == 8, "loss_noalloc_scratch is sized for 8 partial sums per graph");
  double* partials = nullptr;
  DPPM_TRY(noalloc_scratch_ptr(&partials));
  // every one of the B * 8 slots the fold reads is written by the first kernel of this call: nothing has to start at zero
  hipLaunchKernelGGL(loss_bce_kernel, dim3(kLossBlocksPerGraph, B), dim3(256), 0, as_stream(stream), score, target,
                     sigma_dev, mask, N, B_global, dscore, partials);
  DPPM_TRY(check_launch("dppm_loss_bce_fwd_bwd_noalloc"));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fold_partials_kernel<double>), dim3(1), dim3(256), 0, as_stream(stream), (const double*)partials,
                     B * kLossBlocksPerGraph, loss_out);
  return check_launch("dppm_loss_bce_fwd_bwd_noalloc(fold)");
}

extern "C" int dppm_loss_vlb_fwd_bwd(const float* score, const float* A_t, const float* A_0, const double* sched_dev,
                                     const float* mask, int B, int N, float* loss_out, float* dscore, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DPPM_REQUIRE(score && A_t && A_0 && sched_dev && mask && loss_out && workspace, "dppm_loss_vlb_fwd_bwd: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0, "dppm_loss_vlb_fwd_bwd: B=%d N=%d must be positive", B, N);
  DPPM_REQUIRE(workspace_bytes >= dppm_loss_workspace_bytes(B, N), "dppm_loss_vlb_fwd_bwd: workspace %zu < %zu bytes",
               workspace_bytes, dppm_loss_workspace_bytes(B, N));
  double* partials = static_cast<double*>(workspace);
  hipLaunchKernelGGL(loss_vlb_kernel, dim3(kLossBlocksPerGraph, B), dim3(256), 0, as_stream(stream), score, A_t, A_0,
                     sched_dev, mask, N, dscore, partials);
  DPPM_TRY(check_launch("dppm_loss_vlb_fwd_bwd"));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fold_partials_kernel<float>), dim3(1), dim3(256), 0, as_stream(stream), partials,
                     B * kLossBlocksPerGraph, loss_out);
  return check_launch("dppm_loss_vlb_fwd_bwd(fold)");
}
