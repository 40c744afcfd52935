// fp8 (OCP e4m3) operands on the ping-pong schedule (see gemm_kernel.h) + the row-wise quantiser that feeds it.
#include "gemm_kernel.h"

namespace alg {
int launch_gemm_p6_fp8(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s) {
  return launch_gemm<6, 4, true>(a, m_tiles, n_tiles, nwg, s);
}

// one wave per row: amax over the row, then e4m3 conversion of x / scale (v_cvt_pk_fp8_f32 saturates nothing, so clamp)
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const bf16_t* __restrict__ x, int64_t x_rs,
                                                                uint8_t* __restrict__ q, float* __restrict__ scale,
                                                                int64_t rows, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * x_rs;
  float amax = 0.0f;
  for (int c = lane * 8; c < K; c += 512) {
    const uint4 v = *(const uint4*)(xr + c);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      amax = fmaxf(amax, fabsf(__uint_as_float(u[k] << 16)));
      amax = fmaxf(amax, fabsf(__uint_as_float(u[k] & 0xffff0000u)));
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
  const float sc = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + row * (int64_t)K;
  for (int c = lane * 8; c < K; c += 512) {
    const uint4 v = *(const uint4*)(xr + c);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = fminf(fmaxf(__uint_as_float(u[k] << 16) * inv, -448.0f), 448.0f);
      f[2 * k + 1] = fminf(fmaxf(__uint_as_float(u[k] & 0xffff0000u) * inv, -448.0f), 448.0f);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *(uint2*)(qr + c) = make_uint2((unsigned)lo, (unsigned)hi);
  }
}
// The same arithmetic with the whole row in registers (round 5): K = ITERS * 512, every 16-byte load of the row in flight at once, ONE
// pass over the row instead of two (the loop form above reads it for the amax and again for the conversion).  Bit-identical.
template <int ITERS>
__global__ __launch_bounds__(256) void quantize_fp8_rows_reg_kernel(const bf16_t* __restrict__ x, int64_t x_rs,
                                                                    uint8_t* __restrict__ q, float* __restrict__ scale,
                                                                    int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * x_rs + lane * 8;
  uint4 v[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) v[i] = *(const uint4*)(xr + i * 512);
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      amax = fmaxf(amax, fabsf(__uint_as_float(u[k] << 16)));
      amax = fmaxf(amax, fabsf(__uint_as_float(u[k] & 0xffff0000u)));
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
  const float sc = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + row * (int64_t)(ITERS * 512) + lane * 8;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = fminf(fmaxf(__uint_as_float(u[k] << 16) * inv, -448.0f), 448.0f);
      f[2 * k + 1] = fminf(fmaxf(__uint_as_float(u[k] & 0xffff0000u) * inv, -448.0f), 448.0f);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *(uint2*)(qr + i * 512) = make_uint2((unsigned)lo, (unsigned)hi);
  }
}
}  // namespace alg

using namespace alg;

extern "C" int alg_quantize_fp8_rows(const void* x, int64_t x_rstride, void* q, float* scale, int64_t rows, int K,
                                     void* stream) {
  if (rows < 0 || K <= 0 || K % 8 || x_rstride % 8) {
    set_error("alg_quantize_fp8_rows: bad shape rows=%lld K=%d (K %% 8 == 0)", (long long)rows, K);
    return ALG_EINVAL;
  }
  if (rows == 0) return ALG_OK;
  if (!x || !q || !scale || ((uintptr_t)x & 15) || ((uintptr_t)q & 7)) {
    set_error("alg_quantize_fp8_rows: null or misaligned pointer");
    return ALG_EINVAL;
  }
  const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define ALG_QROWS(I)                                                                                                              \
  case I:                                                                                                                         \
    hipLaunchKernelGGL(quantize_fp8_rows_reg_kernel<I>, grid, blk, 0, (hipStream_t)stream, (const bf16_t*)x, x_rstride, (uint8_t*)q, \
                       scale, rows);                                                                                              \
    return check_launch("alg_quantize_fp8_rows");
  if (K % 512 == 0) switch (K / 512) {   // the widths of the Wan blocks (5120, 13824) and their neighbours: the row lives in registers
      ALG_QROWS(6) ALG_QROWS(8) ALG_QROWS(10) ALG_QROWS(12) ALG_QROWS(16) ALG_QROWS(24) ALG_QROWS(27)
      default: break;
    }
#undef ALG_QROWS
  hipLaunchKernelGGL(quantize_fp8_rows_kernel, grid, blk, 0, (hipStream_t)stream,
                     (const bf16_t*)x, x_rstride, (uint8_t*)q, scale, rows, K);
  return check_launch("alg_quantize_fp8_rows");
}
