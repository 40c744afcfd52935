// HBM-bound kernels of the Wan 2.1 VAE (diffusers AutoencoderKLWan; call sites pipeline_wan_image2video_lowpass.py:426-430
// encode of the condition video, :959 decode of the final latents).  The convolutions are launches of the implicit-GEMM
// convolution (alg_conv_cl_bf16) and of alg_gemm_bf16; what is specific to this VAE lives here:
//   * WanRMS_norm (F.normalize over channels * sqrt(C) * gamma) fused with the SiLU that always follows it in the residual
//     blocks, per voxel row of a channels-last buffer whose channel count is padded to the GEMM's power of two;
//   * the row softmax of the mid-block attention (one head of width C = 384 over the H/8 x W/8 tokens of a frame): the
//     scores arrive as TWO bf16 matrices, hi = -bf16(-q k^T) and lo = bf16(q k^T - hi) (the second GEMM takes the first
//     one's output as its residual), so the softmax sees the fp32 scores to 2^-17 instead of bf16-rounded ones.
#include "common.h"

namespace alg {
namespace wanvae {

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(u[k] << 16);
    f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
  }
}

// y[r][c] = act( x[r][c] * sqrt(C) / max(||x[r][:C]||, 1e-12) * gamma[c] ) for c < C, 0 for C <= c < Cp.
// One wave per row; Cp a multiple of 8.  fp32 statistics, one bf16 rounding at the end.
__global__ __launch_bounds__(256) void rms_norm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                            bf16_t* __restrict__ y, int64_t rows, int C, int Cp, int silu) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * Cp;
  float ss = 0.0f;
  for (int c = lane * 8; c < Cp; c += 512) {
    float v[8];
    unpack8(*(const uint4*)(xr + c), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) ss = (c + k < C) ? fmaf(v[k], v[k], ss) : ss;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
  const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
  bf16_t* yr = y + row * Cp;
  for (int c = lane * 8; c < Cp; c += 512) {
    float v[8], g[8], o[8];
    unpack8(*(const uint4*)(xr + c), v);
    unpack8(*(const uint4*)(gamma + c), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = v[k] * inv * g[k];
      if (silu) t = t / (1.0f + __expf(-t));
      o[k] = (c + k < C) ? t : 0.0f;
    }
    uint4 r;
    r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
    *(uint4*)(yr + c) = r;
  }
}

// p[r][j] = softmax_j( (lo[r][j] - neg_hi[r][j]) * scale ) for j < cols, 0 for cols <= j < ld (ld a multiple of 8).
// One workgroup per row; scores are read twice (max, then exp + sum kept in registers for rows up to 256 * 8 * KEEP).
__global__ __launch_bounds__(256) void softmax_hilo_kernel(const bf16_t* __restrict__ neg_hi, const bf16_t* __restrict__ lo,
                                                           bf16_t* __restrict__ p, int cols, int64_t ld, float scale) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const bf16_t* a = neg_hi + row * ld;
  const bf16_t* b = lo + row * ld;
  bf16_t* o = p + row * ld;
  const float c = scale * 1.4426950408889634f;
  float m = -INFINITY;
  for (int j = tid * 8; j < cols; j += 2048) {
    float h[8], l[8];
    unpack8(*(const uint4*)(a + j), h);
    unpack8(*(const uint4*)(b + j), l);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (j + k < cols) m = fmaxf(m, l[k] - h[k]);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (int j = tid * 8; j < cols; j += 2048) {
    float h[8], l[8];
    unpack8(*(const uint4*)(a + j), h);
    unpack8(*(const uint4*)(b + j), l);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (j + k < cols) sum += __builtin_amdgcn_exp2f(((l[k] - h[k]) - m) * c);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int j = tid * 8; j < ld; j += 2048) {
    float h[8], l[8], e[8];
    unpack8(*(const uint4*)(a + j), h);
    unpack8(*(const uint4*)(b + j), l);
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = (j + k < cols) ? __builtin_amdgcn_exp2f(((l[k] - h[k]) - m) * c) * inv : 0.0f;
    uint4 r;
    r.x = pack_bf2(e[0], e[1]); r.y = pack_bf2(e[2], e[3]); r.z = pack_bf2(e[4], e[5]); r.w = pack_bf2(e[6], e[7]);
    *(uint4*)(o + j) = r;
  }
}

}  // namespace wanvae
}  // namespace alg

using namespace alg;

extern "C" int alg_rms_norm_rows(const void* x, const void* gamma, void* y, int64_t rows, int C, int Cp, int silu,
                                 void* stream) {
  if (rows < 0 || C <= 0 || Cp < C || (Cp & 7)) {
    set_error("alg_rms_norm_rows: bad shape rows=%lld C=%d Cp=%d (Cp >= C, Cp %% 8 == 0)", (long long)rows, C, Cp);
    return ALG_EINVAL;
  }
  if (rows == 0) return ALG_OK;
  if (!x || !gamma || !y || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gamma & 15)) {
    set_error("alg_rms_norm_rows: null or misaligned pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(wanvae::rms_norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)gamma, (bf16_t*)y, rows, C, Cp, silu);
  return check_launch("alg_rms_norm_rows");
}

extern "C" int alg_softmax_hilo(const void* neg_hi, const void* lo, void* p, int64_t rows, int cols, int64_t ld, float scale,
                                void* stream) {
  if (rows < 0 || cols <= 0 || ld < cols || (ld & 7) || rows > 0x7fffffff) {
    set_error("alg_softmax_hilo: bad shape rows=%lld cols=%d ld=%lld (ld >= cols, ld %% 8 == 0)", (long long)rows, cols,
              (long long)ld);
    return ALG_EINVAL;
  }
  if (rows == 0) return ALG_OK;
  if (!neg_hi || !lo || !p || ((uintptr_t)neg_hi & 15) || ((uintptr_t)lo & 15) || ((uintptr_t)p & 15)) {
    set_error("alg_softmax_hilo: null or misaligned pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(wanvae::softmax_hilo_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)neg_hi, (const bf16_t*)lo, (bf16_t*)p, cols, ld, scale);
  return check_launch("alg_softmax_hilo");
}
