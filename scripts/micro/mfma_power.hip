// Sustained MFMA throughput and package power with NON-constant operands, 32x32x16 vs 16x16x32 bf16.
// usage: ./mfma_power <seconds> <shape 32|16> <waves_per_simd>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  // 8 different random A and B fragments per lane, rotated so consecutive MFMAs see different operands
  b8 fa[8], fb[8];
  for (int j = 0; j < 8; ++j)
    for (int i = 0; i < 8; ++i) {
      unsigned h = hashu(seed + threadIdx.x * 131u + j * 17u + i);
      fa[j][i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
      fb[j][i] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    }
  if (SHAPE == 32) {
    f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[j], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[(j + 1) & 7], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(j + 1) & 7], fb[j], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(j + 1) & 7], fb[(j + 1) & 7], a3, 0, 0, 0);
      }
    }
    float r = a0[0] + a1[3] + a2[1] + a3[2];
    if (r == 12345.678f) out[0] = r;
  } else {
    f4v a[8];
    for (int j = 0; j < 8; ++j) a[j] = f4v{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(j + q) & 7], fb[(j + (q >> 1)) & 7], a[q], 0, 0, 0);
      }
    }
    float r = 0;
    for (int j = 0; j < 8; ++j) r += a[j][j & 3];
    if (r == 12345.678f) out[0] = r;
  }
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int shape = argc > 2 ? atoi(argv[2]) : 32;
  const int wps = argc > 3 ? atoi(argv[3]) : 2;
  float* d; hipMalloc(&d, 4);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  dim3 grid(pr.multiProcessorCount * wps), blk(256);
  const int iters = 20000;
  // flops per wave per iteration: 32 MFMAs of 32768 flop (shape 32) or 64 MFMAs of 16384 flop (shape 16)
  const double flop_iter = 32.0 * 32768.0;
  auto launch = [&]() {
    if (shape == 32) hipLaunchKernelGGL(k<32>, grid, blk, 0, 0, d, iters, 1234u);
    else hipLaunchKernelGGL(k<16>, grid, blk, 0, 0, d, iters, 1234u);
  };
  launch(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double first = 0, last = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double tf = flop_iter * iters * grid.x * 4 / (ms * 1e-3) / 1e12;
    if (n == 0) first = tf;
    last = tf; ++n;
  }
  printf("MFMA %s bf16 random operands, %d waves/SIMD: first %.0f, sustained %.0f TFLOP/s\n", shape == 32 ? "32x32x16" : "16x16x32", wps, first, last);
  return 0;
}
