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

// gfx950 packed fp32 -> bf16 conversion, round-to-nearest-even: {bf16(hi), bf16(lo)} in one VALU op
// (there is no clang builtin for it; plain asm so the scheduler may move it freely)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
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
