// What the K / V^T pieces cost the default d = 64 attention statement: its per-tile instruction mix alone, with two LDS-DMA pieces per tile and
// wave (the product's form), and with the pieces as plain loads into AccVGPRs + ds_write_b128 one tile later (see gen_attn_dma_cost.py).
// Sustained TFLOP/s under the package power cap, pseudo-random operands.   usage: ./attn_dma_cost <seconds per arm> [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#include "attn_dma_cost.inc"

__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

#define KERNEL(NAME, ASM)                                                                                                  \
  __global__ __launch_bounds__(512) void NAME(float* sink, int iters, unsigned seed, uint64_t* clocks, const char* src) { \
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];   /* 32 KiB read by the fragment reads + 32 KiB the pieces land in */ \
    for (int i = threadIdx.x; i < 16384; i += 512) {                                                                       \
      const unsigned h = hashu(seed + i * 7u + blockIdx.x);                                                                \
      lds[i] = ((h & 0x3fff3fffu) | 0x3c003c00u) ^ (h & 0x80008000u);                                                      \
    }                                                                                                                      \
    __syncthreads();                                                                                                       \
    uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                                                       \
    b8 fb[4];                                                                                                              \
    for (int j = 0; j < 4; ++j)                                                                                            \
      for (int i = 0; i < 8; ++i) {                                                                                        \
        const unsigned h = hashu(seed + (blockIdx.x * 512 + threadIdx.x) * 131u + j * 17u + i);                            \
        fb[j][i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));                                              \
      }                                                                                                                    \
    int n = __builtin_amdgcn_readfirstlane(iters);                                                                         \
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;                               \
    const unsigned addr = base + (threadIdx.x & 63) * 16;                                                                  \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                                     \
    const unsigned wl = __builtin_amdgcn_readfirstlane(base + 32768 + wave * 1024);   /* the wave's 1 KiB of each 8 KiB piece row */ \
    const unsigned wa = base + 32768 + wave * 1024 + (threadIdx.x & 63) * 16;                                              \
    /* each workgroup streams its own 512 KiB window of the (L2-resident) source, 2 KiB per tile and wave */              \
    const uint64_t gb = (uint64_t)(src + (size_t)blockIdx.x * 524288);                                                     \
    unsigned go0 = wave * 1024 + (threadIdx.x & 63) * 16, go1 = go0 + 8192;                                                \
    const unsigned mask = 524287u;                                                                                         \
    const float c = 0.18f, m = -0.25f;                                                                                     \
    asm volatile("v_accvgpr_write_b32 a0, 0\n\t" ASM                                                                      \
                 : [n] "+s"(n), [go0] "+v"(go0), [go1] "+v"(go1)                                                           \
                 : [b0] "v"(fb[0]), [b1] "v"(fb[1]), [b2] "v"(fb[2]), [b3] "v"(fb[3]), [addr] "v"(addr), [c] "s"(c), [m] "v"(m), \
                   [wl] "s"(wl), [wa] "v"(wa), [gb] "s"(gb), [mask] "s"(mask)                                              \
                 : "scc", "memory", "m0", MIX_CLOB);                                                                       \
    if (n == 12345) sink[0] = 1.0f;                                                                                        \
    if (threadIdx.x == 0) {                                                                                                \
      clocks[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;                                                          \
      clocks[blockIdx.x * 2 + 1] = wall_clock64() - r0;                                                                    \
    }                                                                                                                      \
  }
KERNEL(mix_none, MIX_NONE_ASM)
KERNEL(mix_dma, MIX_DMA_ASM)
KERNEL(mix_reg, MIX_REG_ASM)
KERNEL(mix_bar1, MIX_BAR1_ASM)
KERNEL(mix_bar2, MIX_BAR2_ASM)

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;
  float* d;
  uint64_t* clk;
  char* src;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipMalloc(&d, 4);
  hipMalloc(&clk, pr.multiProcessorCount * 16);
  const size_t nsrc = (size_t)pr.multiProcessorCount * 524288 + 65536;
  hipMalloc(&src, nsrc);
  {  // finite bf16 values of magnitude ~1
    unsigned* h = (unsigned*)malloc(nsrc);
    unsigned x = 12345u;
    for (size_t i = 0; i < nsrc / 4; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = ((x & 0x3fff3fffu) | 0x3c003c00u) ^ (x & 0x80008000u);
    }
    hipMemcpy(src, h, nsrc, hipMemcpyHostToDevice);
    free(h);
  }
  const char* names[5] = {"d64 mix alone", "d64 mix + 2 LDS-DMA pieces per tile", "d64 mix + 2 plain loads + 2 ds_write_b128 per tile",
                          "... LDS-DMA + wait and barrier per tile", "... LDS-DMA + wait and barrier per TWO tiles"};
  for (int r = 0; r < rounds; ++r)
    for (int s = 0; s < 5; ++s) {
      const int iters = s >= 3 ? 20000 : 40000;   // the barrier arms run two tiles per trip
      dim3 grid(pr.multiProcessorCount);
      auto launch = [&]() {
        switch (s) {
          case 0: hipLaunchKernelGGL(mix_none, grid, dim3(512), 0, 0, d, iters, 1u + r, clk, src); break;
          case 1: hipLaunchKernelGGL(mix_dma, grid, dim3(512), 0, 0, d, iters, 1u + r, clk, src); break;
          case 2: hipLaunchKernelGGL(mix_reg, grid, dim3(512), 0, 0, d, iters, 1u + r, clk, src); break;
          case 3: hipLaunchKernelGGL(mix_bar1, grid, dim3(512), 0, 0, d, iters, 1u + r, clk, src); break;
          default: hipLaunchKernelGGL(mix_bar2, grid, dim3(512), 0, 0, d, iters, 1u + r, clk, src); break;
        }
      };
      launch();
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      auto t0 = std::chrono::steady_clock::now();
      double last = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipEventRecord(a), launch(), hipEventRecord(b), hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        last = (s >= 3 ? 32 : 16) * 32768.0 * iters * grid.x * 8 / (ms * 1e-3) / 1e12;
        hipEventDestroy(a), hipEventDestroy(b);
      }
      uint64_t h[2];
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double mhz = khz > 0 && h[1] ? (double)h[0] / ((double)h[1] / khz) / 1e3 : 0.0;
      printf("%-52s sustained %6.0f TFLOP/s  clock %5.0f MHz  pipe-busy %.3f\n", names[s], last, mhz,
             mhz > 0 ? last * 1e12 / (pr.multiProcessorCount * 4 * 1024.0 * mhz * 1e6) : 0.0);
      fflush(stdout);
    }
  return 0;
}
