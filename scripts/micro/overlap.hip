// Cross-wave MFMA / VALU overlap on one SIMD (gfx950): a 512-thread workgroup puts waves w and w+4 on the same SIMD.
// role A = MFMA only, role B = VALU only; compare {A alone, B alone, A+B on the same SIMD}.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// mode bit0: low waves (0-3) run MFMA; bit1: high waves (4-7) run VALU (fma); bit2: VALU = exp; bit3: dependent MFMA chain
// bit4: both roles in EVERY wave, MFMA block then VALU block (phase = wave parity flips order)
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float seed) {
  const int wave = threadIdx.x >> 6;
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  f16v acc0 = {0}, acc1 = {0};
  b8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  const bool do_mfma = (mode & 16) ? true : (wave < 4 && (mode & 1));
  const bool do_valu = (mode & 16) ? true : (wave >= 4 && (mode & 2));
  const bool valu_first = (mode & 16) && (wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_valu && valu_first) {
      if (mode & 4) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
      else { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    if (do_mfma) {
      if (mode & 8) { REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));) }
      else { REP4(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                               "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1"
                               : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));) }
    }
    if (do_valu && !valu_first) {
      if (mode & 4) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
      else { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
  }
  float r = a0 + a1 + a2 + a3 + acc0[0] + acc1[3];
  if (r == 12345.678f) out[0] = r;
}

float run(int mode, int blocks_per_cu) {
  float* d; hipMalloc(&d, 4);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  dim3 grid(pr.multiProcessorCount * blocks_per_cu), blk(512);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, grid, blk, 0, 0, d, 10, mode, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, grid, blk, 0, 0, d, 4000, mode, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(d);
  return ms;
}

int main() {
  for (int bpc : {1, 2}) {
    printf("blocks/CU %d (waves/SIMD %d)\n", bpc, 2 * bpc);
    printf("  MFMA only (16/iter, 2 chains)        %.3f ms\n", run(1, bpc));
    printf("  MFMA only (16/iter, 1 dependent chain) %.3f ms\n", run(1 | 8, bpc));
    printf("  fma only  (64/iter)                  %.3f ms\n", run(2, bpc));
    printf("  exp only  (64/iter)                  %.3f ms\n", run(2 | 4, bpc));
    printf("  MFMA wave + fma wave on one SIMD     %.3f ms\n", run(3, bpc));
    printf("  MFMA wave + exp wave on one SIMD     %.3f ms\n", run(3 | 4, bpc));
    printf("  every wave MFMA-block then fma-block, partner wave in opposite order  %.3f ms\n", run(16, bpc));
    printf("  same with exp                        %.3f ms\n", run(16 | 4, bpc));
  }
  return 0;
}
