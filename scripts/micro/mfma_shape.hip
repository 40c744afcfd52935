// Which bf16 MFMA shape is cheaper in ENERGY on gfx950?  Register-only loops on pseudo-random operands under the package power
// cap: sustained TFLOP/s = what the shape delivers per watt (the part lowers its clock until the draw fits).
//   0  v_mfma_f32_32x32x16_bf16, 4 accumulator chains (64 registers)                 -- csrc/calibrate.hip's loop
//   1  v_mfma_f32_16x16x32_bf16, 16 accumulator chains (64 registers)
//   2  v_mfma_f32_32x32x16_bf16, GEMM-like: 4 A x 4 B fragments -> 16 blocks (256 accumulator registers)
//   3  v_mfma_f32_16x16x32_bf16, GEMM-like: 8 A x 8 B fragments -> 64 blocks (256 accumulator registers)
// Per 32768 FLOP the 32x32x16 form reads + writes 1024 fp32 accumulators and 1024 bf16 operands, the 16x16x32 form 512 + 2048:
// less register-file traffic per FLOP if accumulators travel through the file.
// usage: ./mfma_shape <seconds per arm> [rounds]      build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void fill(b8 (&fa)[8], b8 (&fb)[8], unsigned seed) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned h = hashu(seed + (blockIdx.x * 256u + threadIdx.x) * 131u + j * 17u + i);
      fa[j][i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
      fb[j][i] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    }
}
struct Tap {
  uint64_t c0, r0;
  __device__ void start() { c0 = __builtin_readcyclecounter(), r0 = wall_clock64(); }
  __device__ void stop(uint64_t* clocks) {
    if (threadIdx.x == 0) {
      uint64_t* c = clocks + (size_t)blockIdx.x * 2;
      c[0] = __builtin_readcyclecounter() - c0, c[1] = wall_clock64() - r0;
    }
  }
};

#include "mfma_shape.inc"
#define OPS                                                                                                              \
  [a0] "v"(fa[0]), [a1] "v"(fa[1]), [a2] "v"(fa[2]), [a3] "v"(fa[3]), [a4] "v"(fa[4]), [a5] "v"(fa[5]), [a6] "v"(fa[6]),   \
      [a7] "v"(fa[7]), [b0] "v"(fb[0]), [b1] "v"(fb[1]), [b2] "v"(fb[2]), [b3] "v"(fb[3]), [b4] "v"(fb[4]), [b5] "v"(fb[5]), \
      [b6] "v"(fb[6]), [b7] "v"(fb[7])
// every loop trip issues 32 * 32768 FLOP per wave (32 MFMAs of 32x32x16 or 64 of 16x16x32); the accumulators are literal
// AccVGPRs inside one asm statement (nothing the compiler could shuffle)
#define KERNEL(NAME, ASM, CLOB)                                                                        \
  __global__ __launch_bounds__(256) void NAME(float* sink, int iters, unsigned seed, uint64_t* clocks) { \
    Tap t;                                                                                             \
    t.start();                                                                                         \
    b8 fa[8], fb[8];                                                                                   \
    fill(fa, fb, seed);                                                                                \
    int n = __builtin_amdgcn_readfirstlane(iters);                                                     \
    asm volatile(ASM : [n] "+s"(n) : OPS : "scc", CLOB);                                               \
    if (n == 12345) sink[0] = 1.0f;                                                                    \
    t.stop(clocks);                                                                                    \
  }
KERNEL(k0, K0_ASM, CLOB64)
KERNEL(k1, K1_ASM, CLOB64)
KERNEL(k2, K2_ASM, CLOB256)
KERNEL(k3, K3_ASM, CLOB256)

// e4m3 operands (8 registers = 32 bytes per lane) on v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4, block scales 2^0: 4 x the FLOP of
// the bf16 loops per trip.  Random bytes with the NaN codes (0x7f / 0xff) replaced.
typedef int i8v __attribute__((ext_vector_type(8)));
#define KERNEL8(NAME, ASM)                                                                             \
  __global__ __launch_bounds__(256) void NAME(float* sink, int iters, unsigned seed, uint64_t* clocks) { \
    Tap t;                                                                                             \
    t.start();                                                                                         \
    i8v w[4], x[4];                                                                                    \
    for (int j = 0; j < 4; ++j)                                                                        \
      for (int i = 0; i < 8; ++i) {                                                                    \
        unsigned a = hashu(seed + (blockIdx.x * 256u + threadIdx.x) * 131u + j * 17u + i), b = hashu(a + 77u); \
        a &= 0x7e7e7e7eu | 0x80808080u, b &= 0x7e7e7e7eu | 0x80808080u;   /* never exponent + mantissa all ones */ \
        w[j][i] = (int)a, x[j][i] = (int)b;                                                            \
      }                                                                                                \
    int n = __builtin_amdgcn_readfirstlane(iters);                                                     \
    const int sc = 0x7f7f7f7f;                                                                         \
    asm volatile(ASM : [n] "+s"(n)                                                                     \
                 : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), \
                   [x3] "v"(x[3]), [sc] "v"(sc)                                                        \
                 : "scc", CLOB256);                                                                    \
    if (n == 12345) sink[0] = 1.0f;                                                                    \
    t.stop(clocks);                                                                                    \
  }
KERNEL8(k4, K4_ASM)
KERNEL8(k5, K5_ASM)

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;
  float* d;
  uint64_t* clk;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const int iters = 20000;
  const char* names[6] = {"32x32x16, 4 chains", "16x16x32, 16 chains", "32x32x16, 4x4 blocks (256 acc)", "16x16x32, 8x8 blocks (256 acc)",
                          "e4m3 32x32x64 scaled, 256 acc", "e4m3 16x16x128 scaled, 256 acc"};
  for (int wps = 2; wps >= 1; --wps) {
    dim3 grid(pr.multiProcessorCount * wps), blk(256);
    hipMalloc(&d, 4);
    hipMalloc(&clk, grid.x * 16);
    for (int r = 0; r < rounds; ++r)
      for (int s = 0; s < 6; ++s) {
        if (wps == 2 && s >= 2) continue;   // 256 accumulators: one wave per SIMD only
        auto launch = [&]() {
          switch (s) {
            case 0: hipLaunchKernelGGL(k0, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
            case 1: hipLaunchKernelGGL(k1, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
            case 2: hipLaunchKernelGGL(k2, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
            case 3: hipLaunchKernelGGL(k3, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
            case 4: hipLaunchKernelGGL(k4, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
            default: hipLaunchKernelGGL(k5, grid, blk, 0, 0, d, iters, 1u + r, clk); break;
          }
        };
        launch();
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        double first = 0, last = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
          hipEvent_t a, b;
          hipEventCreate(&a), hipEventCreate(&b);
          hipEventRecord(a), launch(), hipEventRecord(b), hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          const double tf = (s >= 4 ? 4.0 : 1.0) * 32.0 * 32768.0 * iters * grid.x * 4 / (ms * 1e-3) / 1e12;
          if (n == 0) first = tf;
          last = tf, ++n;
          hipEventDestroy(a), hipEventDestroy(b);
        }
        uint64_t h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double mhz = khz > 0 && h[1] ? (double)h[0] / ((double)h[1] / khz) / 1e3 : 0.0;
        printf("%d waves/SIMD  %-34s first %6.0f sustained %6.0f TFLOP/s  clock %5.0f MHz  pipe-busy %.3f\n", wps, names[s], first,
               last, mhz, mhz > 0 ? last * 1e12 / (pr.multiProcessorCount * 4 * (s >= 4 ? 2048.0 : 1024.0) * mhz * 1e6) : 0.0);
        fflush(stdout);
      }
    hipFree(d), hipFree(clk);
  }
  return 0;
}
