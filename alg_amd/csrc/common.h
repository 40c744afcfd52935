// Shared device/host helpers for libalg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/alg_hip.h"

namespace alg {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// fp32 value rounded through bf16 (the reference's bf16 tensors round after every op)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

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
