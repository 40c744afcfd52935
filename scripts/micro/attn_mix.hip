// The d = 64 attention instruction mix behind ONE MFMA, fine-interleaved in program order inside a wave: per
// v_mfma_f32_32x32x16_bf16 two v_exp_f32, one v_cvt_pk_bf16_f32, one v_dot2c_f32_bf16 (softmax of one score pair) and one
// ds_read_b128 (the next fragment).  How much of it hides in the MFMA's shadow, at 1 and 2 waves per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o attn_mix attn_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// FILL: 0 none | 1 exp, exp | 2 exp, exp, cvt_pk, dot2c (software-pipelined: cvt / dot2 work on the PREVIOUS pair) |
//       3 = 2 + ds_read_b128 (+ counted lgkmcnt) | 4 = 3 with every second MFMA gap carrying one extra v_add_u32 (address work)
//       5 exp, exp, cvt_pk | 6 exp, exp, cvt_pk, v_add_f32 x 2 (row sum of the unrounded pair, two accumulators) |
//       7 = 6 + ds_read_b128 | 8 exp, exp, cvt_pk, one v_add_f32 + one v_fma-free chain: add(p0, p1) then add to the sum |
//       9 exp, exp, cvt_pk, v_pk_add_f32
template <int FILL, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += NT) ((float*)lds)[i] = seed + i;
  __syncthreads();
  float x0 = seed + threadIdx.x * 1e-3f, x1 = x0 + 0.5f, p0 = 0.f, p1 = 0.f, sum = 0.f, sum1 = 0.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pp = {0.f, 0.f}, ss = {0.f, 0.f};
  unsigned pk = 0, addr = (threadIdx.x & 63) * 16, extra = threadIdx.x;
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  b8 fa, fb, fr;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i + threadIdx.x); fb[i] = (__bf16)(seed - i); fr[i] = fa[i]; }
#define FILLER                                                                                                          \
  if (FILL == 1) asm volatile("v_exp_f32 %0, %2\n v_exp_f32 %1, %3" : "=v"(p0), "=v"(p1) : "v"(x0), "v"(x1));             \
  if (FILL == 5) asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n v_exp_f32 %0, %3\n v_exp_f32 %1, %4" : "+v"(p0), "+v"(p1), "+v"(pk) : "v"(x0), "v"(x1)); \
  if (FILL == 6 || FILL == 7) asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n v_add_f32 %3, %3, %0\n v_add_f32 %4, %4, %1\n v_exp_f32 %0, %5\n v_exp_f32 %1, %6" : "+v"(p0), "+v"(p1), "+v"(pk), "+v"(sum), "+v"(sum1) : "v"(x0), "v"(x1)); \
  if (FILL == 7) asm volatile("s_waitcnt lgkmcnt(3)\n ds_read_b128 %0, %1" : "=v"(fr) : "v"(addr));                          \
  if (FILL == 8) asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n v_add_f32 %4, %0, %1\n v_exp_f32 %0, %5\n v_exp_f32 %1, %6\n v_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(pk), "+v"(sum), "+v"(sum1) : "v"(x0), "v"(x1)); \
  if (FILL == 9) asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n v_pk_add_f32 %3, %3, %6\n v_exp_f32 %0, %4\n v_exp_f32 %1, %5" : "+v"(p0), "+v"(p1), "+v"(pk), "+v"(ss) : "v"(x0), "v"(x1), "v"(pp)); \
  if (FILL >= 2 && FILL <= 4) asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n"                          \
                              "v_dot2c_f32_bf16 %3, 0x3f803f80, %2" : "+v"(p0), "+v"(p1), "+v"(pk), "+v"(sum) : "v"(x0), "v"(x1)); \
  if (FILL == 3 || FILL == 4) asm volatile("s_waitcnt lgkmcnt(3)\n ds_read_b128 %0, %1" : "=v"(fr) : "v"(addr));                          \
  if (FILL == 4) asm volatile("v_add_u32 %0, 16, %0" : "+v"(extra));
#define MF(C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(fa), "v"(fb)); FILLER
  for (int it = 0; it < iters; ++it) {
    REP16(MF(c0) MF(c1) MF(c2) MF(c3))
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float r = p0 + p1 + sum + sum1 + pp[0] + ss[0] + ss[1] + c0[0] + c1[1] + c2[2] + c3[3] + (float)fr[0] + extra + pk;
  if (r == 12345.678f) out[0] = r;
}

template <int FILL, int NT>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 400;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  dim3 grid(pr.multiProcessorCount), blk(NT);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<FILL, NT>), grid, blk, 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<FILL, NT>), grid, blk, 0, 0, d, iters, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // SIMD time per MFMA: NT / 256 waves share a SIMD
  printf("%-72s %6.2f ns per MFMA and wave, %6.2f ns of SIMD time per MFMA\n", name, ms * 1e6 / (iters * 64.0),
         ms * 1e6 / (iters * 64.0) / (NT / 256));
  hipFree(d);
}

int main() {
  run<0, 256>("1 wave/SIMD  MFMA only");
  run<1, 256>("1 wave/SIMD  MFMA + 2 exp");
  run<2, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + dot2c");
  run<3, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + dot2c + ds_read_b128");
  run<4, 256>("1 wave/SIMD  ... + v_add_u32");
  run<0, 512>("2 waves/SIMD MFMA only");
  run<2, 512>("2 waves/SIMD MFMA + 2 exp + cvt_pk + dot2c");
  run<3, 512>("2 waves/SIMD MFMA + 2 exp + cvt_pk + dot2c + ds_read_b128");
  run<4, 512>("2 waves/SIMD ... + v_add_u32");
  run<3, 1024>("4 waves/SIMD MFMA + 2 exp + cvt_pk + dot2c + ds_read_b128");
  run<5, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk");
  run<6, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + 2 v_add_f32");
  run<8, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + add(p0,p1) + add to sum");
  run<9, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + v_pk_add_f32");
  run<7, 256>("1 wave/SIMD  MFMA + 2 exp + cvt_pk + 2 v_add_f32 + ds_read_b128");
  run<6, 512>("2 waves/SIMD MFMA + 2 exp + cvt_pk + 2 v_add_f32");
  run<7, 512>("2 waves/SIMD MFMA + 2 exp + cvt_pk + 2 v_add_f32 + ds_read_b128");
  run<7, 1024>("4 waves/SIMD MFMA + 2 exp + cvt_pk + 2 v_add_f32 + ds_read_b128");
  return 0;
}
