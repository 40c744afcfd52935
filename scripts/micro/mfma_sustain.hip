// Sustained pure-MFMA loop (no memory traffic): what the matrix pipe delivers under the 1400 W package cap.
// usage: ./mfma_sustain <seconds> <waves_per_simd 1|2> <valu_per_mfma 0..7>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  b8 fa, fb;
  float a0 = seed, a1 = seed + 1;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i + threadIdx.x); fb[i] = (__bf16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(fa), "v"(fb));
          if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0));
          if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a1));
          if (NV >= 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a0));
          if (NV >= 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a1));
          asm volatile("v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(fa), "v"(fb));)
  }
  float r = acc0[0] + acc1[3] + acc2[1] + acc3[2] + a0 + a1;
  if (r == 12345.678f) out[0] = r;
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int wps = argc > 2 ? atoi(argv[2]) : 2;
  const int nv = argc > 3 ? atoi(argv[3]) : 0;
  float* d; hipMalloc(&d, 4);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  dim3 grid(pr.multiProcessorCount * wps), blk(256);
  const int iters = 20000;
  auto launch = [&]() {
    if (nv == 0) hipLaunchKernelGGL(k<0>, grid, blk, 0, 0, d, iters, 1.0f);
    else if (nv == 2) hipLaunchKernelGGL(k<2>, grid, blk, 0, 0, d, iters, 1.0f);
    else hipLaunchKernelGGL(k<4>, grid, blk, 0, 0, d, iters, 1.0f);
  };
  launch(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double first = 0, last = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double tf = 2.0 * 32 * 32 * 16 * 32.0 * iters * grid.x * 4 / (ms * 1e-3) / 1e12;
    if (n == 0) first = tf;
    last = tf; ++n;
  }
  printf("pure MFMA 32x32x16 bf16, %d waves/SIMD, %d VALU per 2 MFMA: first %.0f TFLOP/s, sustained %.0f TFLOP/s (%d launches)\n", wps, nv, first, last, n);
  return 0;
}
