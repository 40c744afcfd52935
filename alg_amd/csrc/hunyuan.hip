// HBM-bound kernels specific to the HunyuanVideo DiT forward (diffusers
// HunyuanVideoTransformer3DModel, call site pipeline_hunyuan_video_image2video_lowpass.py:1243-1252): per-head RMSNorm of
// q / k fused with the 3-axis rotary embedding of the latent tokens, the masked mean that pools the prompt for the token
// refiner, SiLU of the conditioning vectors.  Everything else of that forward reuses the CogVideoX / Wan kernels.
#include "common.h"

namespace alg {
namespace hy {

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(u[k] << 16);
    f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
  }
}

// in place on rows of `heads` 128-wide head vectors: x = rope( bf16( bf16(x * rsqrt(mean_head(x^2) + eps)) * w ) ).
// 16 lanes own one head vector (8 elements each); rope (x * cos + rot(x) * sin, interleaved pairs, fp32 tables [.][128])
// only on tokens < rope_tokens of each batch.
__global__ __launch_bounds__(256) void headnorm_rope_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ cos_tab,
                                                            const float* __restrict__ sin_tab, int64_t x_rs,
                                                            int64_t x_bs, int64_t total_rows, int rows, int heads,
                                                            int rope_tokens, float eps) {
  const int64_t vec = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // (row, head) index
  const int sub = threadIdx.x & 15;
  if (vec >= total_rows * heads) return;
  const int64_t row = vec / heads;
  const int head = (int)(vec - row * heads);
  const int tok = (int)(row % rows);
  bf16_t* p = x + (row / rows) * x_bs + (int64_t)tok * x_rs + head * 128 + sub * 8;
  float v[8], wv[8];
  unpack8(*(const uint4*)p, v);
  unpack8(*(const uint4*)(w + sub * 8), wv);
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) q = fmaf(v[k], v[k], q);
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
  const float rstd = rsqrtf(q * (1.0f / 128.0f) + eps);
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = rbf(rbf(v[k] * rstd) * wv[k]);
  if (cos_tab && tok < rope_tokens) {
    const float* c = cos_tab + (int64_t)tok * 128 + sub * 8;
    const float* s = sin_tab + (int64_t)tok * 128 + sub * 8;
    const float4 c0 = *(const float4*)c, c1 = *(const float4*)(c + 4), s0 = *(const float4*)s, s1 = *(const float4*)(s + 4);
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = o[2 * j], b = o[2 * j + 1];
      o[2 * j] = a * cv[2 * j] + (-b) * sv[2 * j];
      o[2 * j + 1] = b * cv[2 * j + 1] + a * sv[2 * j + 1];
    }
  }
  uint4 r;
  r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
  *(uint4*)p = r;
}

// out[b][d] = bf16( sum_{l < valid[b]} x[b][l][d] / valid[b] )   (HunyuanVideoTokenRefiner pooled prompt)
__global__ __launch_bounds__(256) void masked_mean_kernel(const bf16_t* __restrict__ x, const int* __restrict__ valid,
                                                          bf16_t* __restrict__ out, int B, int L, int D) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * D) return;
  const int b = (int)(e / D), d = (int)(e % D);
  const int n = valid[b];
  float acc = 0.0f;
  for (int l = 0; l < n; ++l) acc += bf2f(x[((int64_t)b * L + l) * D + d]);
  out[e] = f2bf(acc / (float)n);
}

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t numel) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = bf2f(x[e]);
    y[e] = f2bf(v / (1.0f + __expf(-v)));
  }
}

}  // namespace hy
}  // namespace alg

using namespace alg;

extern "C" int alg_headnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab,
                                 int64_t x_rstride, int64_t x_bstride, int batch, int rows, int heads, int rope_tokens,
                                 float eps, void* stream) {
  if (batch < 0 || rows < 0 || heads <= 0 || x_rstride % 8 || x_bstride % 8 || x_rstride < (int64_t)heads * 128 ||
      (cos_tab && !sin_tab)) {
    set_error("alg_headnorm_rope: bad shape batch=%d rows=%d heads=%d stride=%lld", batch, rows, heads,
              (long long)x_rstride);
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)batch * rows;
  if (total == 0) return ALG_OK;
  if (!x || !weight) {
    set_error("alg_headnorm_rope: null pointer");
    return ALG_EINVAL;
  }
  const int64_t threads = total * heads * 16;
  hipLaunchKernelGGL(hy::headnorm_rope_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)x, (const bf16_t*)weight, cos_tab, sin_tab, x_rstride, x_bstride, total, rows, heads,
                     rope_tokens, eps);
  return check_launch("alg_headnorm_rope");
}

extern "C" int alg_masked_mean(const void* x, const int* valid, void* out, int batch, int L, int D, void* stream) {
  if (!x || !valid || !out || batch <= 0 || L <= 0 || D <= 0) {
    set_error("alg_masked_mean: bad argument");
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)batch * D;
  hipLaunchKernelGGL(hy::masked_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, valid, (bf16_t*)out, batch, L, D);
  return check_launch("alg_masked_mean");
}

extern "C" int alg_silu(const void* x, void* y, int64_t numel, void* stream) {
  if (numel < 0) {
    set_error("alg_silu: bad argument");
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  if (!x || !y) {
    set_error("alg_silu: null pointer");
    return ALG_EINVAL;
  }
  int64_t want = (numel + 255) / 256;
  hipLaunchKernelGGL(hy::silu_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, numel);
  return check_launch("alg_silu");
}
