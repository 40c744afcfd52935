// Box calibration for the benchmark line: what THIS chip sustains on its matrix pipe right now, and at what
// shader clock -- measured in the run, so that a reader of one bench line can tell a slow box from a slow kernel.
//
//   alg_calib_mfma_bf16   a register-only loop of v_mfma_f32_32x32x16_bf16 on pseudo-random bf16 operands (operand toggling
//                         sets the power draw: constant operands run 15-20 % faster under the 1400 W cap and would flatter the
//                         box), two waves per SIMD, no memory traffic.  The caller brackets it with events on the stream.
//   clock taps            one lane per sampled workgroup reads the shader-cycle counter (s_memtime) and the constant-rate
//                         counter (s_memrealtime) at the start and at the end of its life: d cycles / d wall = the shader clock
//                         that workgroup saw.  The same two reads sit in the product attention kernels behind a NULL-by-default
//                         pointer (alg_attn_clock_tap), so the clock is that of the kernel the roofline line is about, not of a
//                         stand-in.
#include "common.h"

namespace alg {

std::atomic<uint64_t*> g_clock_tap{nullptr};
std::atomic<int> g_clock_tap_slots{0};

__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void calib_mfma_bf16_kernel(float* sink, int iters, unsigned seed, uint64_t* clocks) {
  uint64_t c0 = 0, r0 = 0;
  const bool tap = clocks != nullptr && threadIdx.x == 0;
  if (tap) {
    c0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
  }
  // eight different random A and B fragments per lane, rotated so that consecutive MFMAs see different operands
  bf16x8 fa[8], fb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned h = hashu(seed + (blockIdx.x * 256u + threadIdx.x) * 131u + j * 17u + i);
      fa[j][i] = (short)f2bf(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
      fb[j][i] = (short)f2bf(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    }
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[j], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[(j + 1) & 7], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(j + 1) & 7], fb[j], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(j + 1) & 7], fb[(j + 1) & 7], a3, 0, 0, 0);
    }
  }
  const float r = a0[0] + a1[3] + a2[1] + a3[2];
  if (r == 12345.678f) sink[0] = r;   // keeps the loop alive; never true on these operands
  if (tap) {
    uint64_t* c = clocks + (size_t)blockIdx.x * 4;
    c[0] = c0;
    c[1] = r0;
    c[2] = __builtin_readcyclecounter();
    c[3] = wall_clock64();
  }
}

}  // namespace alg

using namespace alg;

/* One launch of the register-only MFMA loop: `blocks` workgroups of 256 threads (0: two per CU = two waves per SIMD), each wave
 * issuing 32 * iters MFMAs = 32 * iters * 32768 FLOP.  clocks (optional): uint64 [blocks][4] = {cycles, wall} at start / end of
 * each workgroup's lane 0.  Returns the number of workgroups launched (> 0) or a negative ALG_E* code. */
extern "C" int alg_calib_mfma_bf16(float* sink, int iters, unsigned seed, int blocks, uint64_t* clocks, void* stream) {
  if (!sink || iters <= 0 || blocks < 0) {
    set_error("alg_calib_mfma_bf16: bad argument");
    return ALG_EINVAL;
  }
  if (blocks == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      set_error("alg_calib_mfma_bf16: cannot query the device");
      return ALG_ELAUNCH;
    }
    blocks = 2 * cus;
  }
  hipLaunchKernelGGL(calib_mfma_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, seed, clocks);
  const int rc = check_launch("alg_calib_mfma_bf16");
  return rc != ALG_OK ? rc : blocks;
}

/* Rate of the constant counter the clock taps read (s_memrealtime), in kHz; 0 when the runtime cannot tell. */
extern "C" int alg_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

/* Clock tap of the attention kernels (flash_attn_d64_pipe_kernel, flash_attn_d128_q64 family): while `buffer` is non-NULL every
 * launch has lane 0 of each workgroup whose index is a multiple of 64 write {cycles, wall} at its start and end into
 * buffer[(block / 64) % slots][4] (uint64).  Results are unaffected.  NULL (the default) switches it off. */
extern "C" void alg_attn_clock_tap(uint64_t* buffer, int slots) {
  g_clock_tap_slots.store(buffer ? slots : 0, std::memory_order_relaxed);
  g_clock_tap.store(slots > 0 ? buffer : nullptr, std::memory_order_release);
}
