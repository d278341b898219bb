// Graph statistics for model selection on the device: degree and clustering-coefficient
// histograms of sampled adjacencies and the Gaussian-EMD MMD between two sets of histograms, so that sample_testing does
// not stall training on networkx / pyemd loops (reference evaluation/stats.py:41-69, :140-186, evaluation/mmd.py:14-41,
// :90-106).  Integer / double arithmetic: results are bit-exact restatements of the numpy / networkx formulas.
#include "dppm_common.h"

namespace dppm {
namespace {

constexpr int ST_THREADS = 256;

// One workgroup per graph.  Rows of the adjacency as bit masks in LDS (self loops dropped, as adjs_to_graphs does);
// deg_i = popcount(row_i); t_i = sum_{j in nbr(i)} popcount(row_i & row_j) = 2 x triangles through i, which is the `t` of
// networkx's clustering (c_i = t_i / (d_i (d_i - 1)), 0 when t_i = 0).  Isolated nodes are not part of the graph
// (adjs_to_graphs removes them); a graph without edges is the single isolated node adjs_to_graphs adds.
// The clustering bin follows numpy.histogram(range=(0, 1)) exactly: floor(c * bins) corrected against the float64 edges
// numpy.linspace produces (passed in `edges`), last bin closed.
__global__ __launch_bounds__(ST_THREADS) void graph_stats_kernel(const float* __restrict__ adj, int N, int bins,
                                                                 const double* __restrict__ edges, float* __restrict__ deg_hist,
                                                                 float* __restrict__ clus_hist) {
  extern __shared__ unsigned st_lds[];
  const int W = (N + 31) / 32;
  unsigned* rows = st_lds;                                  // [N][W]
  int* dh = reinterpret_cast<int*>(rows + (size_t)N * W);   // [N]
  int* ch = dh + N;                                         // [bins]
  int* nact = ch + bins;                                    // [1]
  const int b = blockIdx.x;
  const float* A = adj + (size_t)b * N * N;
  for (int idx = threadIdx.x; idx < N * W; idx += ST_THREADS) {
    const int i = idx / W, w = idx % W;
    unsigned bits = 0;
    for (int t = 0; t < 32; ++t) {
      const int j = 32 * w + t;
      if (j < N && j != i && A[(size_t)i * N + j] > 0.5f) bits |= 1u << t;
    }
    rows[idx] = bits;
  }
  for (int i = threadIdx.x; i < N; i += ST_THREADS) dh[i] = 0;
  for (int i = threadIdx.x; i < bins; i += ST_THREADS) ch[i] = 0;
  if (threadIdx.x == 0) *nact = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += ST_THREADS) {
    const unsigned* ri = rows + (size_t)i * W;
    int d = 0;
    for (int w = 0; w < W; ++w) d += __popc(ri[w]);
    if (d == 0) continue;
    long t2 = 0;
    for (int w = 0; w < W; ++w) {
      unsigned m = ri[w];
      while (m) {
        const int j = 32 * w + __ffs(m) - 1;
        m &= m - 1;
        const unsigned* rj = rows + (size_t)j * W;
        for (int v = 0; v < W; ++v) t2 += __popc(ri[v] & rj[v]);
      }
    }
    const double c = t2 == 0 ? 0.0 : (double)t2 / ((double)d * (double)(d - 1));
    int k = (int)(c * (double)bins);          // numpy: ((x - first_edge) * (bins / (last - first))).astype(intp)
    if (k == bins) k = bins - 1;
    if (c < edges[k]) --k;
    else if (k != bins - 1 && c >= edges[k + 1]) ++k;
    atomicAdd(&dh[d], 1);
    atomicAdd(&ch[k], 1);
    atomicAdd(nact, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0 && *nact == 0) { dh[0] = 1; ch[0] = 1; }   // the lone node of an edgeless graph: degree 0, clustering 0
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += ST_THREADS) deg_hist[(size_t)b * N + i] = (float)dh[i];
  for (int i = threadIdx.x; i < bins; i += ST_THREADS) clus_hist[(size_t)b * bins + i] = (float)ch[i];
}

// k(x, y) = exp(-emd(x, y)^2 / (2 sigma^2)), emd = sum_k |cdf_x(k) - cdf_y(k)| / distance_scaling for pmfs on a common
// support with ground distance |i - j| (what pyemd returns for the toeplitz distance matrix of evaluation/mmd.py:19-23).
// One thread per ordered pair of the three blocks (1,1), (2,2), (1,2); per-workgroup partial sums, folded in a fixed order.
__global__ __launch_bounds__(ST_THREADS) void mmd_pairs_kernel(const float* __restrict__ h1, int n1, const float* __restrict__ h2, int n2,
                                                               int bins, double sigma, double scaling, double* __restrict__ partial) {
  __shared__ double sh[3][ST_THREADS / 64];
  const long p11 = (long)n1 * n1, p22 = (long)n2 * n2, p12 = (long)n1 * n2;
  const long idx = (long)blockIdx.x * ST_THREADS + threadIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  if (idx < p11 + p22 + p12) {
    int which;
    const float *x, *y;
    if (idx < p11) { which = 0; x = h1 + (idx / n1) * bins; y = h1 + (idx % n1) * bins; }
    else if (idx < p11 + p22) { const long q = idx - p11; which = 1; x = h2 + (q / n2) * bins; y = h2 + (q % n2) * bins; }
    else { const long q = idx - p11 - p22; which = 2; x = h1 + (q / n2) * bins; y = h2 + (q % n2) * bins; }
    double sx = 0.0, sy = 0.0;
    for (int k = 0; k < bins; ++k) { sx += (double)x[k]; sy += (double)y[k]; }
    double cx = 0.0, cy = 0.0, emd = 0.0;
    for (int k = 0; k < bins; ++k) {
      cx += (double)x[k] / sx;
      cy += (double)y[k] / sy;
      emd += fabs(cx - cy);
    }
    emd /= scaling;
    v[which] = exp(-emd * emd / (2.0 * sigma * sigma));
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double s = v[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[c][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int w = 0; w < ST_THREADS / 64; ++w) s += sh[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] = s;
  }
}

__global__ void mmd_fold_kernel(const double* __restrict__ partial, int nblocks, int n1, int n2, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < nblocks; ++b)
      for (int c = 0; c < 3; ++c) s[c] += partial[(size_t)b * 3 + c];
    s[0] /= (double)n1 * n1; s[1] /= (double)n2 * n2; s[2] /= (double)n1 * n2;
    out[0] = s[0] + s[1] - 2.0 * s[2];   // compute_mmd (evaluation/mmd.py:104-106)
    out[1] = s[0]; out[2] = s[1]; out[3] = s[2];
  }
}

inline long mmd_blocks(int n1, int n2) {
  const long pairs = (long)n1 * n1 + (long)n2 * n2 + (long)n1 * n2;
  return (pairs + ST_THREADS - 1) / ST_THREADS;
}

}  // namespace
}  // namespace dppm

using namespace dppm;

extern "C" int dppm_graph_stats(const float* adj, int B, int N, int bins, const double* edges, float* deg_hist, float* clus_hist,
                                void* stream) {
  DPPM_REQUIRE(adj && edges && deg_hist && clus_hist, "dppm_graph_stats: null pointer");
  DPPM_REQUIRE(B > 0 && N > 0 && N <= 1024 && bins > 0 && bins <= 4096, "dppm_graph_stats: B=%d N=%d (<= 1024) bins=%d (<= 4096)", B, N, bins);
  const size_t lds = ((size_t)N * ((N + 31) / 32) + N + bins + 1) * 4;
  DPPM_TRY(lds_limit(reinterpret_cast<const void*>(graph_stats_kernel), (size_t)(lds), "dppm_graph_stats"));
  hipLaunchKernelGGL(graph_stats_kernel, dim3(B), dim3(ST_THREADS), lds, as_stream(stream), adj, N, bins, edges, deg_hist, clus_hist);
  return check_launch("dppm_graph_stats");
}

extern "C" size_t dppm_mmd_workspace_bytes(int n1, int n2) {
  if (n1 <= 0 || n2 <= 0) return 0;
  return (size_t)mmd_blocks(n1, n2) * 3 * sizeof(double);
}

extern "C" int dppm_mmd_gaussian_emd(const float* h1, int n1, const float* h2, int n2, int bins, double sigma, double distance_scaling,
                                     double* out4, void* workspace, size_t workspace_bytes, void* stream) {
  DPPM_REQUIRE(h1 && h2 && out4 && workspace, "dppm_mmd_gaussian_emd: null pointer");
  DPPM_REQUIRE(n1 > 0 && n2 > 0 && bins > 0 && sigma > 0.0 && distance_scaling > 0.0, "dppm_mmd_gaussian_emd: n1=%d n2=%d bins=%d", n1, n2, bins);
  DPPM_REQUIRE(workspace_bytes >= dppm_mmd_workspace_bytes(n1, n2), "dppm_mmd_gaussian_emd: workspace too small");
  const long nb = mmd_blocks(n1, n2);
  DPPM_REQUIRE(nb <= 2147483647L, "dppm_mmd_gaussian_emd: too many pairs");
  double* partial = static_cast<double*>(workspace);
  hipLaunchKernelGGL(mmd_pairs_kernel, dim3((unsigned)nb), dim3(ST_THREADS), 0, as_stream(stream), h1, n1, h2, n2, bins, sigma,
                     distance_scaling, partial);
  DPPM_TRY(check_launch("mmd_pairs"));
  hipLaunchKernelGGL(mmd_fold_kernel, dim3(1), dim3(64), 0, as_stream(stream), partial, (int)nb, n1, n2, out4);
  return check_launch("mmd_fold");
}
