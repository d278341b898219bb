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
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
  return 0;
}
