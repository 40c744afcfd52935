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

typedef unsigned short bf16_t;  // raw bf16 bits

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
