// HBM-bound row kernels of the DiT block: LayerNorm + AdaLN modulation, and per-head QK LayerNorm + RoPE.
// One wave (64 lanes) owns one row / eight head vectors, 16-byte accesses per lane, fp32 statistics with a
// two-pass variance held in registers, cross-lane reduction by DPP-style shuffles (no LDS).
//
// Rounding points mirror the reference's bf16 tensors (every torch op returns bf16): LayerNorm output,
// (1 + scale), the product and the sum are each rounded.
#include "common.h"
#include "qk_norm_rope.h"

namespace alg {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// y = LN(x) * (1 + scale[seg]) + shift[seg]; D = ITERS * 512
template <int ITERS>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                     const bf16_t* __restrict__ w, const bf16_t* __restrict__ bs,
                                                     const bf16_t* __restrict__ scale,
                                                     const bf16_t* __restrict__ shift, int64_t mod_bs,
                                                     int64_t x_bs, int64_t y_bs, int64_t total_rows, int rows,
                                                     int seg_split, int64_t seg_stride, float eps) {
  constexpr int D = ITERS * 512;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int bidx = (int)(row / rows);
  const int r = (int)(row - (int64_t)bidx * rows);
  const int seg = r >= seg_split ? 1 : 0;
  const bf16_t* xr = x + (int64_t)bidx * x_bs + (int64_t)r * D;
  float v[ITERS][8];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    unpack8(*(const uint4*)(xr + i * 512 + lane * 8), v[i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[i][k];
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = v[i][k] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
  const bf16_t* sc = scale ? scale + (int64_t)bidx * mod_bs + (int64_t)seg * seg_stride : nullptr;
  const bf16_t* sh = shift ? shift + (int64_t)bidx * mod_bs + (int64_t)seg * seg_stride : nullptr;
  bf16_t* yr = y + (int64_t)bidx * y_bs + (int64_t)r * D;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c0 = i * 512 + lane * 8;
    float wv[8], bv[8], o[8];
    if (w) unpack8(*(const uint4*)(w + c0), wv);
    if (bs) unpack8(*(const uint4*)(bs + c0), bv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float n = (v[i][k] - mean) * rstd;
      if (w) n = n * wv[k];
      if (bs) n = n + bv[k];
      o[k] = rbf(n);
    }
    if (sc) {
      float scv[8], shv[8];
      unpack8(*(const uint4*)(sc + c0), scv);
      unpack8(*(const uint4*)(sh + c0), shv);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = rbf(rbf(o[k] * rbf(1.0f + scv[k])) + shv[k]);
    }
    *(uint4*)(yr + c0) = pack8(o);
  }
}

// The same arithmetic, rpw consecutive rows per wave (round 4; round 5: rpw chosen by the launcher so that the whole call is ONE
// resident round of waves -- 255 registers = two waves per SIMD = 2048 waves on the chip: with a fixed 8 rows per wave the C2 call was
// 2.17 rounds, and the last sixth of a round cannot keep enough bytes in flight to use the HBM).  ln_mod_kernel re-reads the four parameter rows (weight, bias,
// scale, shift: 4 x 2 D bytes) for every token row -- 30 KB through the vector L1 per 12 KB of HBM traffic at D = 3072, and the
// loads sit behind the statistics.  Here a wave keeps the parameters of its 8 ITERS columns packed in registers across its rows
// (rbf(1 + scale) formed once: the value the per-row form rounds every time), reloads scale / shift only when the (batch item,
// segment) of the next row changes, and has the NEXT row's 16-byte loads in flight while it works on the current one.
// Bit-identical to ln_mod_kernel (same operations in the same order per element).
template <int ITERS>
__global__ __launch_bounds__(256) void ln_mod_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ w, const bf16_t* __restrict__ bs,
                                                          const bf16_t* __restrict__ scale,
                                                          const bf16_t* __restrict__ shift, int64_t mod_bs, int64_t x_bs,
                                                          int64_t y_bs, int64_t total_rows, int rows, int seg_split,
                                                          int64_t seg_stride, float eps, int rpw) {
  constexpr int D = ITERS * 512;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw;
  if (row0 >= total_rows) return;
  const int64_t row_end = row0 + rpw < total_rows ? row0 + rpw : total_rows;
  uint4 wp[ITERS], bp[ITERS], s1p[ITERS], shp[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    wp[i] = *(const uint4*)(w + i * 512 + lane * 8);
    bp[i] = *(const uint4*)(bs + i * 512 + lane * 8);
  }
  auto src = [&](int64_t row) -> const bf16_t* {
    const int bidx = (int)(row / rows);
    return x + (int64_t)bidx * x_bs + (int64_t)(row - (int64_t)bidx * rows) * D + lane * 8;
  };
  uint4 nxt[ITERS];
  {
    const bf16_t* xr = src(row0);
#pragma unroll
    for (int i = 0; i < ITERS; ++i) nxt[i] = *(const uint4*)(xr + i * 512);
  }
  int cur_key = -1;
  for (int64_t row = row0; row < row_end; ++row) {
    float v[ITERS][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      unpack8(nxt[i], v[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[i][k];
    }
    if (row + 1 < row_end) {
      const bf16_t* xr = src(row + 1);
#pragma unroll
      for (int i = 0; i < ITERS; ++i) nxt[i] = *(const uint4*)(xr + i * 512);
    }
    const int bidx = (int)(row / rows);
    const int r = (int)(row - (int64_t)bidx * rows);
    const int key = bidx * 2 + (r >= seg_split ? 1 : 0);
    if (key != cur_key) {   // wave-uniform: the first row of the wave, a new batch item, the text -> video boundary
      cur_key = key;
      const int64_t off = (int64_t)bidx * mod_bs + (int64_t)(key & 1) * seg_stride + lane * 8;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        float scv[8];
        unpack8(*(const uint4*)(scale + off + i * 512), scv);
#pragma unroll
        for (int k = 0; k < 8; ++k) scv[k] = 1.0f + scv[k];
        s1p[i] = pack8(scv);                                  // = rbf(1 + scale), packed
        shp[i] = *(const uint4*)(shift + off + i * 512);
      }
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = v[i][k] - mean;
        q = fmaf(d, d, q);
      }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    bf16_t* yr = y + (int64_t)bidx * y_bs + (int64_t)r * D + lane * 8;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      float wv[8], bv[8], s1[8], shv[8], o[8];
      unpack8(wp[i], wv);
      unpack8(bp[i], bv);
      unpack8(s1p[i], s1);
      unpack8(shp[i], shv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float n = (v[i][k] - mean) * rstd;
        n = n * wv[k];
        n = n + bv[k];
        o[k] = rbf(rbf(rbf(n) * s1[k]) + shv[k]);
      }
      *(uint4*)(yr + i * 512) = pack8(o);
    }
  }
}

// In place per-head LayerNorm(64) + RoPE on qk [batch][S][2][heads][64]; 8 lanes per head vector.
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ qk, const bf16_t* __restrict__ wq,
                                                           const bf16_t* __restrict__ bq,
                                                           const bf16_t* __restrict__ wk,
                                                           const bf16_t* __restrict__ bk,
                                                           const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, int64_t total_vec, int S,
                                                           int heads, int text_len, float eps, float q_scale) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int64_t vec = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
  const bool live = vec < total_vec;
  const int64_t vv = live ? vec : total_vec - 1;
  const int hv = (int)(vv % (2 * heads));
  const int64_t tok = vv / (2 * heads);
  const int s = (int)(tok % S);
  const bool is_k = hv >= heads;
  bf16_t* ptr = qk + vv * 64 + sub * 8;
  float v[8];
  unpack8(*(const uint4*)ptr, v);
  float sum = qk_chunk_sum(v);
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  sum += __shfl_xor(sum, 4, 64);
  const float mean = qk_mean(sum);
  float q = qk_chunk_sqdev(v, mean);
  q += __shfl_xor(q, 1, 64);
  q += __shfl_xor(q, 2, 64);
  q += __shfl_xor(q, 4, 64);
  const float rstd = qk_rstd(q, eps);
  float wv[8], bv[8];
  unpack8(*(const uint4*)((is_k ? wk : wq) + sub * 8), wv);
  unpack8(*(const uint4*)((is_k ? bk : bq) + sub * 8), bv);
  float o[8];
  // q_scale (alg_qk_norm_rope_scaled): the softmax scale * log2(e) folded into Q where it is produced, inside the LAST
  // rounding of the row (after the rope for video tokens, in the LayerNorm rounding for text tokens): the attention kernel
  // then needs no per-score multiply.  K is never scaled.  (Arithmetic: qk_norm_rope.h, shared with the GEMM store loop.)
  const float qs = is_k ? 1.0f : q_scale;
  const bool roped = s >= text_len && cos_tab;
  qk_ln_chunk(v, mean, rstd, wv, bv, roped, qs, o);
  if (roped) {
    const int64_t pos = (int64_t)(s - text_len) * 64 + sub * 8;
    const float4 c0 = *(const float4*)(cos_tab + pos), c1 = *(const float4*)(cos_tab + pos + 4);
    const float4 s0 = *(const float4*)(sin_tab + pos), s1 = *(const float4*)(sin_tab + pos + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    qk_rope_chunk(o, cs, sn, qs);
  }
  if (live) *(uint4*)ptr = pack8(o);
}

// The same arithmetic, one wave per TOKEN (round 5): the 2 * heads head vectors of a token share their rotary row, so the 64 cos + 64
// sin floats a lane group needs (16 per lane) are loaded ONCE per token instead of once per head vector -- in the per-vector kernel above
// every 16 bytes of qk came with 64 bytes of table through the vector L1.  Four head-vector loads in flight per lane.  Needs
// 2 * heads % 32 == 0 (CogVideoX: 96); bit-identical to qk_norm_rope_kernel (same helpers, same order per element).
__global__ __launch_bounds__(256) void qk_norm_rope_token_kernel(bf16_t* __restrict__ qk, const bf16_t* __restrict__ wq,
                                                                 const bf16_t* __restrict__ bq, const bf16_t* __restrict__ wk,
                                                                 const bf16_t* __restrict__ bk, const float* __restrict__ cos_tab,
                                                                 const float* __restrict__ sin_tab, int64_t tokens, int S, int heads,
                                                                 int text_len, float eps, float q_scale) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7, slot = lane >> 3;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const int s = (int)(tok % S);
  const bool roped = s >= text_len && cos_tab;
  float cs[8], sn[8];
  if (roped) {
    const int64_t pos = (int64_t)(s - text_len) * 64 + sub * 8;
    const float4 c0 = *(const float4*)(cos_tab + pos), c1 = *(const float4*)(cos_tab + pos + 4);
    const float4 s0 = *(const float4*)(sin_tab + pos), s1 = *(const float4*)(sin_tab + pos + 4);
    cs[0] = c0.x, cs[1] = c0.y, cs[2] = c0.z, cs[3] = c0.w, cs[4] = c1.x, cs[5] = c1.y, cs[6] = c1.z, cs[7] = c1.w;
    sn[0] = s0.x, sn[1] = s0.y, sn[2] = s0.z, sn[3] = s0.w, sn[4] = s1.x, sn[5] = s1.y, sn[6] = s1.z, sn[7] = s1.w;
  }
  float wqv[8], bqv[8], wkv[8], bkv[8];
  unpack8(*(const uint4*)(wq + sub * 8), wqv);
  unpack8(*(const uint4*)(bq + sub * 8), bqv);
  unpack8(*(const uint4*)(wk + sub * 8), wkv);
  unpack8(*(const uint4*)(bk + sub * 8), bkv);
  bf16_t* const base = qk + tok * (int64_t)(2 * heads) * 64 + sub * 8;
  const int passes = 2 * heads / 8;
  for (int p0 = 0; p0 < passes; p0 += 4) {
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(base + (int64_t)((p0 + u) * 8 + slot) * 64);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hv = (p0 + u) * 8 + slot;
      const bool is_k = hv >= heads;
      float v[8];
      unpack8(raw[u], v);
      float sum = qk_chunk_sum(v);
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      const float mean = qk_mean(sum);
      float q = qk_chunk_sqdev(v, mean);
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      q += __shfl_xor(q, 4, 64);
      const float rstd = qk_rstd(q, eps);
      const float qs = is_k ? 1.0f : q_scale;
      float o[8];
      qk_ln_chunk(v, mean, rstd, is_k ? wkv : wqv, is_k ? bkv : bqv, roped, qs, o);
      if (roped) qk_rope_chunk(o, cs, sn, qs);
      *(uint4*)(base + (int64_t)hv * 64) = pack8(o);
    }
  }
}

}  // namespace alg

using namespace alg;

extern "C" int alg_layernorm_modulate(const void* x, void* y, const void* weight, const void* bias,
                                      const void* scale, const void* shift, int64_t mod_bstride, int batch, int rows,
                                      int D, int64_t x_bstride, int64_t y_bstride, int seg_split, float eps,
                                      void* stream) {
  return alg_layernorm_modulate_seg(x, y, weight, bias, scale, shift, mod_bstride, D, batch, rows, D, x_bstride, y_bstride,
                                    seg_split, eps, stream);
}

extern "C" int alg_layernorm_modulate_seg(const void* x, void* y, const void* weight, const void* bias,
                                          const void* scale, const void* shift, int64_t mod_bstride,
                                          int64_t seg_stride, int batch, int rows, int D, int64_t x_bstride,
                                          int64_t y_bstride, int seg_split, float eps, void* stream) {
  if (!x || !y || batch <= 0 || rows <= 0 || D <= 0 || seg_stride % 8) {
    set_error("alg_layernorm_modulate: bad argument (batch=%d rows=%d D=%d)", batch, rows, D);
    return ALG_EINVAL;
  }
  if (D % 512 != 0 || D > 8192) {
    set_error("alg_layernorm_modulate: D=%d must be a multiple of 512 and <= 8192", D);
    return ALG_EINVAL;
  }
  if ((scale == nullptr) != (shift == nullptr)) {
    set_error("alg_layernorm_modulate: scale and shift must be given together");
    return ALG_EINVAL;
  }
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) || (mod_bstride % 8) || (x_bstride % 8) ||
      (y_bstride % 8)) {
    set_error("alg_layernorm_modulate: pointers must be 16-byte aligned");
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)batch * rows;
  const unsigned grid = (unsigned)((total + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  // many rows with all four parameter rows present (every AdaLN of the CogVideoX / HunyuanVideo blocks): the multi-row form
  if (weight && bias && scale && D <= 3072 && total >= 4096) {
    const int64_t resident_waves = (int64_t)device_cus() * 8;   // CUs x 4 SIMDs x 2 waves of this kernel (2048 on an MI355X)
    const int rpw = (int)std::max<int64_t>(8, (total + resident_waves - 1) / resident_waves);
    const unsigned g2 = (unsigned)((total + 4 * (int64_t)rpw - 1) / (4 * (int64_t)rpw));
#define LN_ROWS(I)                                                                                                      \
  case I:                                                                                                               \
    hipLaunchKernelGGL((ln_mod_rows_kernel<I>), dim3(g2), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,                 \
                       (const bf16_t*)weight, (const bf16_t*)bias, (const bf16_t*)scale, (const bf16_t*)shift,          \
                       mod_bstride, x_bstride, y_bstride, total, rows, seg_split, seg_stride, eps, rpw);                \
    break;
    switch (D / 512) { LN_ROWS(1) LN_ROWS(2) LN_ROWS(3) LN_ROWS(4) LN_ROWS(5) LN_ROWS(6) }
#undef LN_ROWS
    return check_launch("alg_layernorm_modulate");
  }
#define LN_CASE(I)                                                                                                  \
  case I:                                                                                                           \
    hipLaunchKernelGGL(ln_mod_kernel<I>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,                  \
                       (const bf16_t*)weight, (const bf16_t*)bias, (const bf16_t*)scale, (const bf16_t*)shift,      \
                       mod_bstride, x_bstride, y_bstride, total, rows, seg_split, seg_stride, eps);                                                   \
    break;
  switch (D / 512) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    LN_CASE(9) LN_CASE(10) LN_CASE(11) LN_CASE(12) LN_CASE(13) LN_CASE(14) LN_CASE(15) LN_CASE(16)
  }
#undef LN_CASE
  return check_launch("alg_layernorm_modulate");
}

extern "C" int alg_qk_norm_rope_scaled(void* qk, const void* wq, const void* bq, const void* wk, const void* bk,
                                       const float* cos_tab, const float* sin_tab, int batch, int S, int heads,
                                       int text_len, float eps, float q_scale, void* stream) {
  if (!qk || !wq || !bq || !wk || !bk || batch <= 0 || S <= 0 || heads <= 0 || text_len < 0) {
    set_error("alg_qk_norm_rope: bad argument (batch=%d S=%d heads=%d text_len=%d)", batch, S, heads, text_len);
    return ALG_EINVAL;
  }
  if ((cos_tab == nullptr) != (sin_tab == nullptr)) {
    set_error("alg_qk_norm_rope: cos and sin tables must be given together");
    return ALG_EINVAL;
  }
  if (((uintptr_t)qk & 15) || ((uintptr_t)wq & 15) || ((uintptr_t)bq & 15) || ((uintptr_t)wk & 15) ||
      ((uintptr_t)bk & 15) || ((uintptr_t)cos_tab & 15) || ((uintptr_t)sin_tab & 15)) {
    set_error("alg_qk_norm_rope: pointers must be 16-byte aligned");
    return ALG_EINVAL;
  }
  const int64_t total_vec = (int64_t)batch * S * 2 * heads;
  if ((2 * heads) % 32 == 0 && ((int64_t)batch * S + 3) / 4 <= 0x7fffffff) {   // one wave per token: the rotary row is loaded once
    const int64_t tokens = (int64_t)batch * S;
    hipLaunchKernelGGL(qk_norm_rope_token_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qk,
                       (const bf16_t*)wq, (const bf16_t*)bq, (const bf16_t*)wk, (const bf16_t*)bk, cos_tab, sin_tab, tokens, S, heads,
                       text_len, eps, q_scale);
    return check_launch("alg_qk_norm_rope");
  }
  const unsigned grid = (unsigned)((total_vec + 31) / 32);
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qk,
                     (const bf16_t*)wq, (const bf16_t*)bq, (const bf16_t*)wk, (const bf16_t*)bk, cos_tab, sin_tab,
                     total_vec, S, heads, text_len, eps, q_scale);
  return check_launch("alg_qk_norm_rope");
}

extern "C" int alg_qk_norm_rope(void* qk, const void* wq, const void* bq, const void* wk, const void* bk, const float* cos_tab,
                                const float* sin_tab, int batch, int S, int heads, int text_len, float eps, void* stream) {
  return alg_qk_norm_rope_scaled(qk, wq, bq, wk, bk, cos_tab, sin_tab, batch, S, heads, text_len, eps, 1.0f, stream);
}
