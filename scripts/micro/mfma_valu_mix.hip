// One wave per SIMD: how much of a short VALU group hides behind an MFMA, by operand register file and filler kind.
// Group = 1 MFMA 32x32x16 bf16 (4 independent accumulators round-robin) + the filler.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// MODE bits: 1 = srcA/srcB in AccVGPRs, 2 = accumulators in AccVGPRs, FILL: 0 none, 1 fma+exp (dependent), 2 fma+exp+cvt,
// 3 = fma+exp + s_waitcnt lgkmcnt(2) + s_nop, 4 = two independent fma, 5 = fma + exp independent
template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  b8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i + threadIdx.x); fb[i] = (__bf16)(seed - i); }
#define FILLER                                                                                         \
  if (FILL == 1) asm volatile("v_fma_f32 %0, %1, %1, %2\n v_exp_f32 %0, %0" : "+v"(a0) : "v"(a1), "v"(a2));            \
  if (FILL == 2) asm volatile("v_fma_f32 %0, %1, %1, %2\n v_exp_f32 %0, %0\n v_cvt_pk_bf16_f32 %3, %0, %1" : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3)); \
  if (FILL == 3) asm volatile("s_waitcnt lgkmcnt(2)\n v_fma_f32 %0, %1, %1, %2\n v_exp_f32 %0, %0\n s_nop 0" : "+v"(a0) : "v"(a1), "v"(a2)); \
  if (FILL == 4) asm volatile("v_fma_f32 %0, %1, %1, %2\n v_fma_f32 %3, %1, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3));  \
  if (FILL == 5) asm volatile("v_fma_f32 %0, %1, %1, %2\n v_exp_f32 %3, %1" : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3));
#define MF(C)                                                                                                          \
  if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(fa), "v"(fb));                 \
  if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C) : "a"(fa), "a"(fb));                 \
  if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(fa), "v"(fb));                 \
  if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C) : "a"(fa), "a"(fb));                 \
  FILLER
  for (int it = 0; it < iters; ++it) {
    REP16(MF(c0) MF(c1) MF(c2) MF(c3))
  }
  float r = a0 + a3 + c0[0] + c1[1] + c2[2] + c3[3];
  if (r == 12345.678f) out[0] = r;
}

template <int MODE, int FILL>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 500;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  dim3 grid(pr.multiProcessorCount), blk(256);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, FILL>), grid, blk, 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, FILL>), grid, blk, 0, 0, d, iters, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-64s %.2f ns per MFMA group\n", name, ms * 1e6 / (iters * 64.0));
  hipFree(d);
}

int main() {
  run<0, 0>("MFMA only, A/B/C in VGPRs");
  run<1, 0>("MFMA only, A/B in AGPRs, C in VGPRs");
  run<2, 0>("MFMA only, A/B in VGPRs, C in AGPRs");
  run<3, 0>("MFMA only, all in AGPRs");
  run<0, 1>("VGPR MFMA + fma -> exp (dependent)");
  run<1, 1>("A/B AGPR, C VGPR + fma -> exp");
  run<2, 1>("A/B VGPR, C AGPR + fma -> exp");
  run<3, 1>("all AGPR + fma -> exp");
  run<0, 5>("VGPR MFMA + fma, exp (independent)");
  run<0, 4>("VGPR MFMA + 2 fma");
  run<0, 2>("VGPR MFMA + fma -> exp -> cvt_pk");
  run<1, 2>("A/B AGPR, C VGPR + fma -> exp -> cvt_pk");
  run<0, 3>("VGPR MFMA + s_waitcnt lgkmcnt(2), fma -> exp, s_nop 0");
  run<1, 3>("A/B AGPR, C VGPR + s_waitcnt lgkmcnt(2), fma -> exp, s_nop 0");
  return 0;
}
