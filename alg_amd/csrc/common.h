// Shared device/host helpers for libalg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <stdio.h>

#include "../../include/alg_hip.h"

namespace alg {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Run-time options (capi.hip).  The environment is read ONCE, when the library is loaded; launch paths read an int from a
// table and never call getenv.  A host that changes a variable afterwards calls alg_reload_env() (tests do).  Every option of
// the default build selects between schedules that are bit-identical or documented equivalents (the names are in
// capi.hip's table and in README.md).
enum Opt {
  OPT_ATTN_SPLIT_TAIL,  // ALG_ATTN_SPLIT_TAIL   1 (default) | 0: the d = 64 attention as a single launch (no split-KV tail)
  OPT_ATTN_PP,          // ALG_ATTN_PP           4 (8-wave pipelined main launch, v_mfma_f32_32x32x16_bf16) | 7: the same construction on
                        //                       v_mfma_f32_16x16x32_bf16 (attention64_m16.hip; a call it declines runs the default, 4) |
                        //                       0: the straight loop
  OPT_ATTN_VARIANT,     // ALG_ATTN_VARIANT      33 (default: lazy running max) | 1: exact running max, fp32 row sums
  OPT_ATTN128_PIPE,     // ALG_ATTN128_PIPE      1 (default: pipelined d = 128 kernel) | 0: the straight loop
  OPT_ATTN128_Q64,      // ALG_ATTN128_Q64       1 (default: 64-queries-per-wave kernel for >= 4,096 keys) | 2: for every call it can
                        //                       take (>= 512 keys) | 3: as 2, statement off (the frame's C++ tile body only: tests) | 0: off
  OPT_GEMM_PIPE,        // ALG_GEMM_PIPE         10 (default since round 6: the asm main loop on v_mfma_f32_16x16x32_bf16) | 9: the asm main
                        //                       loop on 32x32x16 | 6: the 8-wave ping-pong schedule (9 and 6 are bit-identical; 10 sums 32
                        //                       products per instruction: other fp32 rounding points, same error bound)
  OPT_LOWPASS_PATH,     // ALG_LOWPASS_PATH      0 auto | 1 plane-per-workgroup | 2 lowpass_v2 | 3 lowpass_v3 at any plane
                        //                       count | 4 global-memory passes (all bit-identical)
  OPT_COUNT
};
int opt(Opt o);

// One-time opt-in to more than 64 KiB of dynamic LDS, remembered PER DEVICE (hipFuncSetAttribute is a per-device property: a
// process that drives several GPUs must set it on each).  Racing first calls both set it -- idempotent.
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
};
inline int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;   // unknown: never cached, set on every call
  return dev;
}
inline bool device_done(const PerDeviceOnce& o, int slot) {
  return slot >= 0 && ((o.mask.load(std::memory_order_acquire) >> slot) & 1ull);
}
inline void device_mark(PerDeviceOnce& o, int slot) {
  if (slot >= 0) o.mask.fetch_or(1ull << slot, std::memory_order_release);
}

// Compute units of the current device (256 on an MI355X; fewer in a partitioned mode or on another SKU), cached per device: the
// launchers that size a grid to ONE resident round of waves derive it from this, never from a constant.
inline int device_cus() {
  static std::atomic<int> cache[64];
  const int slot = current_device_slot();
  if (slot >= 0) {
    const int c = cache[slot].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 256;   // unknown: the MI355X figure (sizing only -- never correctness)
  if (slot >= 0) cache[slot].store(cus, std::memory_order_relaxed);
  return cus;
}

// Clock tap of the attention kernels (calibrate.hip: alg_attn_clock_tap).  A launch takes the tap only while one is installed AND
// its stream is not capturing: a device pointer baked into a hipGraph would outlive the buffer it points to.
extern std::atomic<uint64_t*> g_clock_tap;
extern std::atomic<int> g_clock_tap_slots;
inline uint64_t* clock_tap_for(hipStream_t s, int* slots) {
  *slots = 0;
  uint64_t* c = g_clock_tap.load(std::memory_order_acquire);
  const int n = c ? g_clock_tap_slots.load(std::memory_order_relaxed) : 0;
  if (!c || n <= 0) return nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  *slots = n;
  return c;
}

typedef unsigned short bf16_t;  // raw bf16 bits

// Lazy running max of the attention kernels: probabilities are formed against the CURRENT offset and the exact path (tile max,
// grow the offset, rescale O and l) runs only when a tile's row sum leaves [0, 2^80).  fp32 and bf16 keep their relative
// precision at any magnitude, so the limit only has to keep everything finite: at most 2^11 tiles of at most 2^80 each per row
// (l, O <= 2^97 |v|), and 1 / l stays a normal number.  Round 4 raised it from 2^40: with scores ~ N(0, 8^2) log2 units a wave
// crossed 2^40 somewhere along a 17,776-key row in a quarter of all cases, left the pipelined statement for good and held its
// whole workgroup to the C++ loop's pace (867 vs 1130 TFLOP/s for the same launch on N(0, 1.44^2) scores).  The generated
// statements carry the same constant (0x67800000; scripts/gen_attn*_pipe*.py).
#define ALG_LAZY_SUM_LIMIT 1.2089258196146292e24f   /* 2^80 */

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// gfx950 packed fp32 -> bf16 conversion, round-to-nearest-even: {bf16(hi), bf16(lo)} in one VALU op.
// Written as a vector fptrunc so hipcc selects v_cvt_pk_bf16_f32 itself AND pads the VALU->MFMA operand hazard;
// the same instruction in inline asm fed an MFMA B operand without wait states and gave wrong P@V products.
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2_native v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native));
}

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, f) & 0xffffu); }

// fp32 value rounded through bf16 (the reference's bf16 tensors round after every op)
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(pack_bf2(f, f) << 16); }

template <typename T>
__device__ __forceinline__ float load_as_float(const T* p, int64_t i);
template <>
__device__ __forceinline__ float load_as_float<float>(const float* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float load_as_float<bf16_t>(const bf16_t* p, int64_t i) { return bf2f(p[i]); }

template <typename T>
__device__ __forceinline__ void store_from_float(T* p, int64_t i, float v);
template <>
__device__ __forceinline__ void store_from_float<float>(float* p, int64_t i, float v) { p[i] = v; }
template <>
__device__ __forceinline__ void store_from_float<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }

}  // namespace alg
