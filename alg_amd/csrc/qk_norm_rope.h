// Per-head QK LayerNorm(64) + rotary embedding of the CogVideoX attention (diffusers CogVideoXAttnProcessor2_0: norm_q / norm_k, then
// apply_rotary_emb on the video tokens), written ONCE for its two users: the stand-alone in-place kernel (norm.hip) and the
// store loop of the Q|K projection (gemm_kernel.h, alg_gemm_bf16_pair_qk).  Both must give the same bits, so every rounding
// point is spelled out and contraction is off inside these functions whatever the including file is compiled with
// (norm.hip: -ffp-contract=off, the GEMM files: fast).
//
// A head vector is 64 bf16 = eight 16-byte chunks.  Reduction order (fixed): a chunk's eight values are summed left to right,
// chunks combine as ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7)) -- what three xor-shuffles over eight lanes give, and what
// two xor-shuffles over four lanes followed by one add of the two half-vectors give.
#pragma once
#include "common.h"

namespace alg {

struct QkNormRope {           // device-side copy of alg_qk_norm_rope_args
  const bf16_t *wq, *bq, *wk, *bk;
  const float *cos_tab, *sin_tab;
  int heads, text_len;
  float eps, q_scale;
};

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(u[k] << 16);
    f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float qk_chunk_sum(const float (&v)[8]) {
#pragma clang fp contract(off)
  float sum = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) sum += v[k];
  return sum;
}

__device__ __forceinline__ float qk_chunk_sqdev(const float (&v)[8], float mean) {
#pragma clang fp contract(off)
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float d = v[k] - mean;
    q = fmaf(d, d, q);
  }
  return q;
}

__device__ __forceinline__ float qk_mean(float sum64) {
#pragma clang fp contract(off)
  return sum64 * (1.0f / 64);
}

__device__ __forceinline__ float qk_rstd(float sq64, float eps) {
#pragma clang fp contract(off)
  const float var = sq64 * (1.0f / 64);
  return rsqrtf(var + eps);
}

// LayerNorm affine of one chunk, rounded to bf16 as norm_q / norm_k return it.  `qs` (softmax scale * log2 e folded into Q,
// alg_qk_norm_rope_scaled) enters this rounding only for rows that are NOT rotated afterwards; rotated rows take it in the
// rope's rounding.
__device__ __forceinline__ void qk_ln_chunk(const float (&v)[8], float mean, float rstd, const float (&wv)[8],
                                            const float (&bv)[8], bool roped, float qs, float (&o)[8]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float c = v[k] - mean;
    const float n = c * rstd;
    const float a = n * wv[k];
    const float ln = a + bv[k];
    o[k] = roped ? rbf(ln) : (qs == 1.0f ? rbf(ln) : rbf(ln * qs));
  }
}

// x.float() * cos + rotate(x).float() * sin, rotate = (-x_odd, x_even) interleaved; unfused like eager.  Results are left in
// fp32 for the caller's ONE rounding (pack8).
__device__ __forceinline__ void qk_rope_chunk(float (&o)[8], const float (&cs)[8], const float (&sn)[8], float qs) {
#pragma clang fp contract(off)
  float r[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a0 = o[2 * k] * cs[2 * k];
    const float b0 = (-o[2 * k + 1]) * sn[2 * k];
    const float a1 = o[2 * k + 1] * cs[2 * k + 1];
    const float b1 = o[2 * k] * sn[2 * k + 1];
    r[2 * k] = a0 + b0;
    r[2 * k + 1] = a1 + b1;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = qs == 1.0f ? r[k] : r[k] * qs;
}

}  // namespace alg
