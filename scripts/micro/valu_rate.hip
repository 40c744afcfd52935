// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of a few opcodes at full occupancy.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  typedef float f16v __attribute__((ext_vector_type(16)));
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  f16v acc0 = {0}, acc1 = {0};
  b8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {  // v_fma_f32 x8 independent
      REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 1) {  // v_exp_f32
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 2) {  // v_pk_fma_f32
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (OP == 3) {  // v_max3_f32
      REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 4) {  // v_cvt_pk_bf16_f32
      REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 5) {  // mfma 32x32x16 bf16, two independent chains
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                         "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1"
                         : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));)
    } else if (OP == 6) {  // mfma + 7 fma behind each (co-issue test)
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 7) {  // mfma + 2 exp behind each
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 10) {  // mfma + 7 v_pk_fma behind each: does packed fp32 share the matrix datapath?
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n"
                         "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (OP == 11) {  // mfma + 7 v_max3
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %4\n v_max3_f32 %7, %7, %4, %5\n"
                         "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %4\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 12) {  // mfma + 7 v_cvt_pk_bf16_f32
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %4\n"
                         "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 8) {  // v_pk_add_f32
      REP16(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (OP == 20) {  // v_exp_legacy_f32
      REP16(asm volatile("v_exp_legacy_f32 %0, %0\n v_exp_legacy_f32 %1, %1\n v_exp_legacy_f32 %2, %2\n v_exp_legacy_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 21) {  // v_exp_f16
      REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 22) {  // v_dot2c_f32_bf16
      REP16(asm volatile("v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %1, %2, %3\n v_dot2c_f32_bf16 %2, %3, %0\n v_dot2c_f32_bf16 %3, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 23) {  // v_ldexp_f32
      REP16(asm volatile("v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 24) {  // mfma + 7 v_dot2c behind each
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_dot2c_f32_bf16 %4, %5, %6\n v_dot2c_f32_bf16 %5, %6, %7\n v_dot2c_f32_bf16 %6, %7, %4\n v_dot2c_f32_bf16 %7, %4, %5\n"
                         "v_dot2c_f32_bf16 %4, %5, %6\n v_dot2c_f32_bf16 %5, %6, %7\n v_dot2c_f32_bf16 %6, %7, %4\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 25) {  // v_rcp_f32 (another transcendental, for reference)
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 9) {  // v_cndmask
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    }
  }
  float r = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + acc0[0] + acc1[3] + (float)fa[0];
  if (r == 12345.678f) out[0] = r;
}

template <int OP>
void run(const char* name, int per_iter, int waves_per_simd) {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 2000;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  dim3 grid(cus * waves_per_simd), blk(256);  // 4 waves per block -> one per SIMD; waves_per_simd blocks per CU
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, grid, blk, 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, grid, blk, 0, 0, d, iters, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // each SIMD executed waves_per_simd * iters * per_iter instructions
  const double n = (double)waves_per_simd * iters * per_iter;
  printf("%-28s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms,
         ms * 1e6 / n);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", 64, w);
    run<1>("v_exp_f32", 64, w);
    run<20>("v_exp_legacy_f32", 64, w);
    run<21>("v_exp_f16", 64, w);
    run<25>("v_rcp_f32", 64, w);
    run<22>("v_dot2c_f32_bf16", 64, w);
    run<23>("v_ldexp_f32", 64, w);
    run<24>("mfma + 7 dot2c (per group)", 16, w);
    run<2>("v_pk_fma_f32", 64, w);
    run<8>("v_pk_add_f32", 64, w);
    run<3>("v_max3_f32", 64, w);
    run<4>("v_cvt_pk_bf16_f32", 64, w);
    run<9>("v_cndmask_b32", 64, w);
    run<5>("mfma_32x32x16_bf16", 64, w);
    run<6>("mfma + 7 fma (per group)", 16, w);
    run<7>("mfma + 2 exp (per group)", 16, w);
    run<10>("mfma + 7 pk_fma (per group)", 16, w);
    run<11>("mfma + 7 max3 (per group)", 16, w);
    run<12>("mfma + 7 cvt_pk_bf16 (per group)", 16, w);
  }
  return 0;
}
