// Hardware probe: v_pk_mul_f32 / v_pk_fma_f32 whose second source takes the HIGH register of its 64-bit pair for both lanes
// (op_sel:[0,1]) against the same arithmetic with scalar VALU ops, in the instruction neighbourhood of w4_mlp_kernel's in_lin stagers:
// right behind a ds_write2st64_b64 that still has to read the destination pair, operands fresh from a global load, other waves of the
// workgroup busy with LDS reads and MFMAs.  MODE 0: op_sel:[0,1] behind the DS store; 1: op_sel:[0,1] without the store;
// 2: op_sel_hi:[1,0] (low register for both lanes) behind the store; 3: as 0 with one s_nop between the store and the multiply.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(768) void probe(const float4* __restrict__ in, unsigned long long* bad, int rounds) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (wave >= 4) {   // the multipliers' side: LDS fragment reads + MFMAs
    const char* base = lds + 65536 + lane * 16;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds * 6; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uintx4 v = *reinterpret_cast<const volatile uintx4*>(base + ((k * 1024 + r * 8192) & 65535));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
      }
    }
    if (acc[0] == 12345.678f) bad[1] = 1;
    return;
  }
  unsigned long long nbad = 0;
  const floatx2 w0 = {0.37f + 0.001f * tid, -0.52f + 0.002f * tid}, w1 = {0.11f - 0.003f * tid, 0.77f + 0.0005f * tid}, bias = {0.05f * (tid & 7), -0.03f * (tid & 15)};
  floatx2 accv = {1.f, 2.f};
  unsigned long long d0 = 0x1111222233334444ull;
  for (int r = 0; r < rounds; ++r) {
    const float4 A = in[(size_t)(r & 1023) * 256 + (tid & 255)], S = in[(size_t)((r + 517) & 1023) * 256 + (tid & 255)];   // global loads, as the stagers' tile inputs
    floatx2 a01 = {A.x, A.y}, s01 = {S.x, S.y};
    unsigned addr = tid * 8, addr2 = tid * 8 + 16;
    floatx2 acc = accv;
    if (MODE == 0)
      asm volatile("ds_write2st64_b64 %0, %2, %1 offset1:32\n\tv_pk_mul_f32 %1, %3, %5 op_sel:[0,1]\n\tv_add_u32 %0, 0, %7\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,1,0]\n\tv_add_u32 %0, 0x100, %0\n\tv_pk_add_f32 %1, %8, %1"
                   : "+v"(addr), "+v"(acc) : "v"(d0), "v"(w0), "v"(w1), "v"(a01), "v"(s01), "v"(addr2), "v"(bias) : "memory");
    else if (MODE == 1)
      asm volatile("v_pk_mul_f32 %1, %3, %5 op_sel:[0,1]\n\tv_add_u32 %0, 0, %7\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,1,0]\n\tv_add_u32 %0, 0x100, %0\n\tv_pk_add_f32 %1, %8, %1"
                   : "+v"(addr), "+v"(acc) : "v"(d0), "v"(w0), "v"(w1), "v"(a01), "v"(s01), "v"(addr2), "v"(bias) : "memory");
    else if (MODE == 2)
      asm volatile("ds_write2st64_b64 %0, %2, %1 offset1:32\n\tv_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n\tv_add_u32 %0, 0, %7\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel_hi:[1,0,1]\n\tv_add_u32 %0, 0x100, %0\n\tv_pk_add_f32 %1, %8, %1"
                   : "+v"(addr), "+v"(acc) : "v"(d0), "v"(w0), "v"(w1), "v"(a01), "v"(s01), "v"(addr2), "v"(bias) : "memory");
    else
      asm volatile("ds_write2st64_b64 %0, %2, %1 offset1:32\n\ts_nop 0\n\tv_pk_mul_f32 %1, %3, %5 op_sel:[0,1]\n\tv_add_u32 %0, 0, %7\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,1,0]\n\tv_add_u32 %0, 0x100, %0\n\tv_pk_add_f32 %1, %8, %1"
                   : "+v"(addr), "+v"(acc) : "v"(d0), "v"(w0), "v"(w1), "v"(a01), "v"(s01), "v"(addr2), "v"(bias) : "memory");
    const float m = (MODE == 2) ? A.x : A.y, sg = (MODE == 2) ? S.x : S.y;
    const float e0 = __fadd_rn(bias[0], __fmaf_rn(w1[0], sg, __fmul_rn(w0[0], m))), e1 = __fadd_rn(bias[1], __fmaf_rn(w1[1], sg, __fmul_rn(w0[1], m)));
    nbad += (acc[0] != e0) + (acc[1] != e1);
    accv = acc; d0 += addr;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int MODE> static void run(const char* what, const float4* din, unsigned long long* dbad, int rounds) {
  (void)hipMemset(dbad, 0, 32);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(768), 131072, 0, din, dbad, rounds);
  unsigned long long h = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
  printf("%-86s wrong results %llu of %llu\n", what, h, 2ull * rounds * 256 * 256);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long* dbad; float4* din;
  (void)hipMalloc(&dbad, 32);
  const size_t n = 1024 * 256;
  float4* h = (float4*)malloc(n * sizeof(float4));
  unsigned s = 12345;
  for (size_t i = 0; i < n; ++i) { float v[4]; for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; v[k] = ((s >> 28) == 0) ? 1.f : (((s >> 20) & 63) == 0 ? 0.2041f : 0.f); } h[i] = make_float4(v[0], v[1], v[2], v[3]); }
  (void)hipMalloc(&din, n * sizeof(float4)); (void)hipMemcpy(din, h, n * sizeof(float4), hipMemcpyHostToDevice);
  run<0>("op_sel:[0,1] (high register for both lanes) right behind ds_write2st64_b64 of the pair", din, dbad, rounds);
  run<1>("op_sel:[0,1] without the DS store", din, dbad, rounds);
  run<2>("op_sel_hi:[1,0] (low register for both lanes) behind the DS store", din, dbad, rounds);
  run<3>("op_sel:[0,1] behind the DS store with one s_nop between", din, dbad, rounds);
  return 0;
}
