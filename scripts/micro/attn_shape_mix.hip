// The attention statements' per-tile instruction mixes on both bf16 MFMA shapes (see gen_attn_shape_mix.py): sustained TFLOP/s under
// the package power cap, pseudo-random operands.   usage: ./attn_shape_mix <seconds per arm> [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#include "attn_shape_mix.inc"

__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

#define KERNEL(NAME, ASM, NT, CLOB)                                                                          \
  __global__ __launch_bounds__(NT) void NAME(float* sink, int iters, unsigned seed, uint64_t* clocks) { \
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];                                        \
    for (int i = threadIdx.x; i < 8192; i += NT) {                                                     \
      const unsigned h = hashu(seed + i * 7u + blockIdx.x);                                            \
      lds[i] = (h & 0x3fff3fffu) | 0x3c003c00u;   /* bf16 pairs of magnitude ~1 with random mantissas and signs */ \
      lds[i] ^= (h & 0x80008000u);                                                                     \
    }                                                                                                  \
    __syncthreads();                                                                                   \
    uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                                   \
    b8 fb[4];                                                                                          \
    for (int j = 0; j < 4; ++j)                                                                        \
      for (int i = 0; i < 8; ++i) {                                                                    \
        const unsigned h = hashu(seed + (blockIdx.x * NT + threadIdx.x) * 131u + j * 17u + i);         \
        fb[j][i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));                          \
      }                                                                                                \
    int n = __builtin_amdgcn_readfirstlane(iters);                                                     \
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x & 63) * 16; \
    const float c = 0.18f, m = -0.25f;                                                                 \
    asm volatile("v_accvgpr_write_b32 a0, 0\n\t" ASM : [n] "+s"(n)                                    \
                 : [b0] "v"(fb[0]), [b1] "v"(fb[1]), [b2] "v"(fb[2]), [b3] "v"(fb[3]), [addr] "v"(addr), [c] "s"(c), [m] "v"(m) \
                 : "scc", "memory", CLOB);                                                         \
    if (n == 12345) sink[0] = 1.0f;                                                                    \
    if (threadIdx.x == 0) {                                                                            \
      clocks[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;                                      \
      clocks[blockIdx.x * 2 + 1] = wall_clock64() - r0;                                                \
    }                                                                                                  \
  }
KERNEL(d64_32, D64_32_ASM, 512, MIX_CLOB64)
KERNEL(d64_16, D64_16_ASM, 512, MIX_CLOB64)
KERNEL(d128_32, D128_32_ASM, 256, MIX_CLOB128)
KERNEL(d128_16, D128_16_ASM, 256, MIX_CLOB128)

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;
  float* d;
  uint64_t* clk;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipMalloc(&d, 4);
  hipMalloc(&clk, pr.multiProcessorCount * 16);
  const char* names[4] = {"d64 mix, 32x32x16 (2 waves/SIMD)", "d64 mix, 16x16x32 (2 waves/SIMD)", "d128 mix, 32x32x16 (1 wave/SIMD)",
                          "d128 mix, 16x16x32 (1 wave/SIMD)"};
  const double mfma_per_trip[4] = {16, 16, 64, 64};   // in 32x32x16 equivalents (32768 FLOP)
  const int waves[4] = {8, 8, 4, 4};
  for (int r = 0; r < rounds; ++r)
    for (int s = 0; s < 4; ++s) {
      const int iters = s < 2 ? 40000 : 10000;
      dim3 grid(pr.multiProcessorCount);
      auto launch = [&]() {
        switch (s) {
          case 0: hipLaunchKernelGGL(d64_32, grid, dim3(512), 0, 0, d, iters, 1u + r, clk); break;
          case 1: hipLaunchKernelGGL(d64_16, grid, dim3(512), 0, 0, d, iters, 1u + r, clk); break;
          case 2: hipLaunchKernelGGL(d128_32, grid, dim3(256), 0, 0, d, iters, 1u + r, clk); break;
          default: hipLaunchKernelGGL(d128_16, grid, dim3(256), 0, 0, d, iters, 1u + r, clk); break;
        }
      };
      launch();
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      int n = 0;
      double last = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipEventRecord(a), launch(), hipEventRecord(b), hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        last = mfma_per_trip[s] * 32768.0 * iters * grid.x * waves[s] / (ms * 1e-3) / 1e12, ++n;
        hipEventDestroy(a), hipEventDestroy(b);
      }
      uint64_t h[2];
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double mhz = khz > 0 && h[1] ? (double)h[0] / ((double)h[1] / khz) / 1e3 : 0.0;
      printf("%-40s sustained %6.0f TFLOP/s  clock %5.0f MHz  pipe-busy %.3f\n", names[s], last, mhz,
             mhz > 0 ? last * 1e12 / (pr.multiProcessorCount * 4 * 1024.0 * mhz * 1e6) : 0.0);
      fflush(stdout);
    }
  return 0;
}
