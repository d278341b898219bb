// Hardware probe: is a VALU write to the SECOND data operand of ds_write2st64_b64, issued right behind the store, a hazard?
// (hipcc's hazard recogniser pads VALU writes behind DS stores whose FIRST data operand is wider than 64 bits -- ds_write_b96 / b128 --
// and nothing behind ds_write2_b64 / ds_write2st64_b64, which carry 2 x 64 bits of data as well.)
// Waves 0-3 of a 16-wave workgroup store a known pattern with the instruction under test and overwrite a source register GAP
// instructions later; waves 4-15 keep the LDS busy with ds_read_b128 streams so that the store's source read can be delayed.
// After every round the pattern is read back and compared.  Build: hipcc --offload-arch=gfx950 -O2 -o ds_write2_hazard ds_write2_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: overwrite data1 at +1 (v_mov_b64); 1: data1 at +1 with v_pk_mul_f32; 2: data0 at +1; 3: data1 at +3 (two independent VALU between); 4: address at +1; 5: data1 at +2
__global__ __launch_bounds__(1024) void probe(unsigned long long* bad, int rounds) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned long long nbad = 0;
  if (wave >= 4) {   // LDS load: conflict-free 16-byte reads over the upper 64 KB
    const char* base = lds + 65536 + (tid & 63) * 16;
    uintx4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds * 24; ++r) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uintx4 v = *reinterpret_cast<const volatile uintx4*>(base + ((k * 1024 + r * 16384) & 65535));
        acc += v;
      }
    }
    if (acc[0] == 0x12345678u) bad[1] = acc[1];   // keep the reads alive
    return;
  }
  const unsigned addr = tid * 8;   // 256 threads x 8 bytes = 2 KB per image; second image 32 * 64 * 8 = 16 KB further
  for (int r = 0; r < rounds; ++r) {
    unsigned long long d0 = ((unsigned long long)(r * 2654435761u + tid) << 32) | (unsigned)(tid * 40503u + r);
    unsigned long long d1 = ~d0 ^ 0x5555aaaa5555aaaaull;
    const unsigned long long e0 = d0, e1 = d1;
    unsigned long long poison = 0xdeadbeefdeadbeefull, t0 = r, t1 = tid;
    unsigned a = addr;
    if (MODE == 0)
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_mov_b64 %2, %3" : "+v"(a), "+v"(d0), "+v"(d1) : "v"(poison) : "memory");
    else if (MODE == 1)
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_pk_mul_f32 %2, %3, %3" : "+v"(a), "+v"(d0), "+v"(d1) : "v"(poison) : "memory");
    else if (MODE == 2)
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_mov_b64 %1, %3" : "+v"(a), "+v"(d0), "+v"(d1) : "v"(poison) : "memory");
    else if (MODE == 3)
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_mov_b64 %4, %3\n\tv_mov_b64 %5, %3\n\tv_mov_b64 %2, %3" : "+v"(a), "+v"(d0), "+v"(d1), "+v"(poison), "+v"(t0), "+v"(t1) : : "memory");
    else if (MODE == 4)
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_mov_b32 %0, 0" : "+v"(a), "+v"(d0), "+v"(d1) : "v"(poison) : "memory");
    else
      asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32\n\tv_mov_b64 %4, %3\n\tv_mov_b64 %2, %3" : "+v"(a), "+v"(d0), "+v"(d1), "+v"(poison), "+v"(t0) : : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long g0 = *reinterpret_cast<const volatile unsigned long long*>(lds + addr);
    const unsigned long long g1 = *reinterpret_cast<const volatile unsigned long long*>(lds + addr + 16384);
    nbad += (g0 != e0) + (g1 != e1);
    if (t0 == 0x123456789ull && t1 == 77) bad[2] = d0 + d1 + a;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int MODE> static void run(const char* what, unsigned long long* dbad, int rounds) {
  hipMemset(dbad, 0, 32);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(1024), 131072, 0, dbad, rounds);
  unsigned long long h = 0;
  hipDeviceSynchronize();
  hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
  printf("%-70s mismatches %llu of %llu stores\n", what, h, 2ull * rounds * 256 * 256);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long* dbad;
  hipMalloc(&dbad, 32);
  run<2>("data0 overwritten by the next instruction (v_mov_b64)", dbad, rounds);
  run<0>("data1 overwritten by the next instruction (v_mov_b64)", dbad, rounds);
  run<1>("data1 overwritten by the next instruction (v_pk_mul_f32)", dbad, rounds);
  run<5>("data1 overwritten two instructions later", dbad, rounds);
  run<3>("data1 overwritten three instructions later", dbad, rounds);
  run<4>("address overwritten by the next instruction", dbad, rounds);
  hipFree(dbad);
  return 0;
}
