// Philox4x32-10 (Salmon et al., SC'11) for the counter-based Bernoulli draws.  The numpy twin used by the tests is
// oracle/philox.py; both are checked against the Random123 known-answer vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dppm {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

// u(seed, graph, offset, i, j) in [0,1): 24 random bits, exactly representable in float.
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t graph_id, uint32_t offset, uint32_t i, uint32_t j) {
  const uint4 r = philox4x32_10(make_uint4(i, j, graph_id, offset), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  return (float)(r.x >> 8) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace dppm
