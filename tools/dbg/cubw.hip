// debug: what one MI355X moves per second across the CU boundary with the access shapes of the streaming kernels here
// (256 persistent workgroups, 512 loading threads each, float4 per lane, 12 loads in flight per thread):
//   hbm_read   every workgroup streams its own chunk of a 4 GiB buffer
//   l2_read    the 32 workgroups of an XCD keep re-reading one 2 MiB region (L2 hits)
//   write      every workgroup writes its own chunk
//   copy       read own chunk, write own chunk
//   pair       two workgroups of an XCD read the SAME chunk (second reader: L2) and each writes its own: the MLP pair's shape
// build: hipcc -O3 --offload-arch=gfx950 tools/dbg/cubw.hip -o tools/dbg/cubw ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 512;
constexpr int INFLIGHT = 12;

// mode bits: 1 read, 2 write; rd_chunk / wr_chunk: float4 index of the workgroup's chunk; wrap: float4 count of the region that is
// walked over and over (a power of two; n4 for one pass over the chunk)
__global__ __launch_bounds__(THREADS) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, size_t wrap,
                                                         int mode, int share, float* __restrict__ sink) {
  const int wg = blockIdx.x;
  const int xcd = wg & 7, slot = wg >> 3;                    // workgroups go to XCDs round-robin
  size_t rbase;
  if (share == 0) rbase = (size_t)wg * n4;                   // own chunk
  else if (share == 1) rbase = (size_t)xcd * wrap;           // one region per XCD
  else rbase = (size_t)(xcd + 8 * (slot >> 1)) * n4;         // pairs (slot, slot ^ 1) of an XCD share a chunk
  const size_t wbase = (size_t)wg * n4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t start = share == 1 ? ((size_t)slot * 4096) & (wrap - 1) : 0;
  for (size_t i = threadIdx.x; i < n4; i += (size_t)THREADS * INFLIGHT) {
    float4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const size_t k = i + (size_t)u * THREADS;
      const size_t kk = k < n4 ? k : threadIdx.x;
      if (mode & 1) v[u] = src[rbase + (share == 1 ? (start + kk) & (wrap - 1) : kk)];
      else v[u] = make_float4((float)kk, 1.f, 2.f, 3.f);
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const size_t k = i + (size_t)u * THREADS;
      if (mode & 2) { if (k < n4) dst[wbase + k] = v[u]; }
      else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (!(mode & 2) && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[wg] = acc.x;   // keeps the loads alive
}

// The MLP pair's access shape: X [B][128][P] read in tiles of 64 positions x 128 channel rows (256-byte row pieces, row stride
// 4 P bytes), by the workgroups of job 0 and job 1 (neighbours on one XCD take the same tile); each job writes its own Y in the
// same shape.  Thread mapping of the staging waves: rows 4 (tid >> 4) .. + 3, float4 column tid & 15; three tiles in flight.
// narrow: stores in the multipliers' mapping instead (a wave's store covers 16 rows x 64 bytes).
__global__ __launch_bounds__(THREADS) void tile_kernel(const float* __restrict__ X, float* __restrict__ Y0, float* __restrict__ Y1, int P, int B,
                                                       int mode, int njobs, int narrow, float* __restrict__ sink) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int xcd = wg & 7, slot = wg >> 3;
  const int job = njobs == 2 ? (slot & 1) : 0;
  const int rank = njobs == 2 ? xcd + 8 * (slot >> 1) : wg, per_job = 256 / njobs;
  const int tpp = P / 64, ntiles = B * tpp;
  float* Y = job ? Y1 : Y0;
  const size_t ro = (size_t)(4 * (tid >> 4)) * P + 4 * (tid & 15);
  const int lane = tid & 63, wave = tid >> 6;
  const size_t no = (size_t)(16 * wave + (lane & 15)) * P + 4 * (lane >> 4);   // narrow: row 16 wave + ci, positions 4 g + 16 j
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = rank; i < ntiles; i += 3 * per_job) {
    float4 v[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = i + u * per_job < ntiles ? i + u * per_job : i;
      const int b = t / tpp, p0 = (t - b * tpp) * 64;
      const float* Xt = X + (size_t)b * 128 * P + p0 + ro;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (mode & 1) v[u][r] = *reinterpret_cast<const float4*>(Xt + (size_t)r * P);
        else v[u][r] = make_float4((float)t, 1.f, 2.f, 3.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = i + u * per_job;
      if (t >= ntiles) break;
      const int b = t / tpp, p0 = (t - b * tpp) * 64;
      if (mode & 2) {
        float* Yt = Y + (size_t)b * 128 * P + p0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (narrow) *reinterpret_cast<float4*>(Yt + no + 16 * r) = v[u][r];
          else *reinterpret_cast<float4*>(Yt + ro + (size_t)r * P) = v[u][r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc.x += v[u][r].x; acc.y += v[u][r].y; acc.z += v[u][r].z; acc.w += v[u][r].w; }
      }
    }
  }
  if (!(mode & 2) && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[wg] = acc.x;
}

static void run_tile(const char* name, const float* X, float* Y0, float* Y1, int P, int B, int mode, int njobs, int narrow, float* sink, double units) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(tile_kernel, dim3(256), dim3(THREADS), 0, 0, X, Y0, Y1, P, B, mode, njobs, narrow, sink);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tile_kernel, dim3(256), dim3(THREADS), 0, 0, X, Y0, Y1, P, B, mode, njobs, narrow, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double unit = (double)B * 128 * P * 4;
  printf("%-22s %8.3f ms  %4.1f plane-sets of %.0f MB across the CU boundary: %6.2f TB/s\n", name, ms, units, unit / 1e6, units * unit / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

static double run(const char* name, const float4* src, float4* dst, size_t n4, size_t wrap, int mode, int share, float* sink, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(THREADS), 0, 0, src, dst, n4, wrap, mode, share, sink);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(THREADS), 0, 0, src, dst, n4, wrap, mode, share, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-10s %8.3f ms  %6.2f TB/s across the CU boundary\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  fflush(stdout);
  return ms;
}

int main() {
  const size_t chunk_bytes = (size_t)8 << 20;                // 8 MiB per workgroup: 2 GiB per buffer
  const size_t n4 = chunk_bytes / 16;
  const size_t total = 256 * chunk_bytes;
  float4 *a, *b; float* sink;
  CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&sink, 256 * sizeof(float)));
  CK(hipMemset(a, 0, total)); CK(hipMemset(b, 0, total));
  run("hbm_read", a, b, n4, n4, 1, 0, sink, (double)total);
  run("l2_read", a, b, n4, ((size_t)2 << 20) / 16, 1, 1, sink, (double)total);
  run("l2_read_1M", a, b, n4, ((size_t)1 << 20) / 16, 1, 1, sink, (double)total);
  run("write", a, b, n4, n4, 2, 0, sink, (double)total);
  run("copy", a, b, n4, n4, 3, 0, sink, 2.0 * total);
  run("pair", a, b, n4, n4, 3, 2, sink, 2.0 * total);
  {
    const int P = 40000, B = 32;                              // N = 200, 32 graphs: the bench geometry
    const float* X = reinterpret_cast<const float*>(a);
    float* Y0 = reinterpret_cast<float*>(b);
    float* Y1 = Y0 + (size_t)B * 128 * P;                     // 3 x 655 MB < 2 GiB per buffer
    run_tile("tile read, 1 job", X, Y0, Y1, P, B, 1, 1, 0, sink, 1);
    run_tile("tile read, 2 jobs", X, Y0, Y1, P, B, 1, 2, 0, sink, 2);
    run_tile("tile write, 2 jobs", X, Y0, Y1, P, B, 2, 2, 0, sink, 2);
    run_tile("tile write narrow", X, Y0, Y1, P, B, 2, 2, 1, sink, 2);
    run_tile("tile mlp shape", X, Y0, Y1, P, B, 3, 2, 0, sink, 4);
    run_tile("tile mlp shape narrow", X, Y0, Y1, P, B, 3, 2, 1, sink, 4);
    run_tile("tile copy, 1 job", X, Y0, Y1, P, B, 3, 1, 0, sink, 2);
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
  return 0;
}
