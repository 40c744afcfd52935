// Text-encoder front-end kernels: transformers' T5EncoderModel (cog:228-268) and UMT5EncoderModel
// (wan:185-234) around the bf16 GEMM.  Sequences are a few hundred tokens, once per video: these are small VALU / LDS
// kernels, nothing here is worth an MFMA tile.
//   alg_embed_rows      token embedding gather
//   alg_t5_layernorm    T5LayerNorm: bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w), out of place
//   alg_attn_bias       eager attention with an additive relative-position bias and a key mask, head_dim 64 / 80, every tensor
//                       op rounded as the eager bf16 graph does (scores, + bias, fp32 softmax -> bf16 P, P @ V)
//   alg_mul_bf16        gated-GELU product
#include "common.h"

namespace alg {
namespace t5 {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                    bf16_t* __restrict__ out, int64_t n, int D, int vocab) {
  const int chunks = D >> 3;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * chunks; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / chunks;
    const int c = (int)(e - r * chunks);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *(uint4*)(out + r * D + c * 8) = *(const uint4*)(table + id * D + c * 8);
  }
}

// one wave per row
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        bf16_t* __restrict__ y, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * D;
  float q = 0.0f;
  for (int c = lane; c < D; c += 64) {
    const float v = bf2f(xr[c]);
    q = fmaf(v, v, q);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  for (int c = lane; c < D; c += 64) y[row * D + c] = f2bf(rbf(bf2f(xr[c]) * rstd) * bf2f(w[c]));
}

// grid (row blocks, batch * heads); 4 waves, wave w takes query rows row0 + w, + 4, ...; DH = head_dim (64: T5 / UMT5,
// 80: CLIP ViT-H)
// LDS: KT [DH][Lp] bf16 | V [Lp][DH] bf16 | q [4][128] f32 | p [4][Lp] f32
template <int DH>
__global__ __launch_bounds__(256) void attn_bias_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                        const bf16_t* __restrict__ bias_table,
                                                        const int* __restrict__ rel_bucket, const int* __restrict__ mask,
                                                        int heads, int L, int Lp, int64_t qkv_rs, int64_t out_rs,
                                                        int rows_per_wg, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* KT = (bf16_t*)smem;
  bf16_t* V = KT + (size_t)DH * Lp;
  float* qs = (float*)(V + (size_t)Lp * DH);
  float* ps = qs + 4 * 128;
  constexpr int CH = DH / 8;  // 16-byte chunks per head row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y / heads, h = blockIdx.y % heads;
  const bf16_t* kb = k + (int64_t)b * L * qkv_rs + h * DH;
  const bf16_t* vb = v + (int64_t)b * L * qkv_rs + h * DH;
  for (int e = tid; e < Lp * CH; e += 256) {
    const int j = e / CH, c = (e - j * CH) * 8;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (j < L) {
      kv = *(const uint4*)(kb + (int64_t)j * qkv_rs + c);
      vv = *(const uint4*)(vb + (int64_t)j * qkv_rs + c);
    }
    *(uint4*)(V + (size_t)j * DH + c) = vv;
    const uint32_t u[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      KT[(size_t)(c + 2 * t) * Lp + j] = (bf16_t)(u[t] & 0xffffu);
      KT[(size_t)(c + 2 * t + 1) * Lp + j] = (bf16_t)(u[t] >> 16);
    }
  }
  __syncthreads();
  const int nchunk = Lp >> 6;  // <= 8
  const int row0 = blockIdx.x * rows_per_wg;
  const int row1 = min(row0 + rows_per_wg, L);
  const int* mb = mask ? mask + (int64_t)b * L : nullptr;
  float* qw = qs + wave * 128;
  float* pw = ps + (size_t)wave * Lp;
  for (int i = row0 + wave; i < row1; i += 4) {
    const bf16_t* qr = q + ((int64_t)b * L + i) * qkv_rs + h * DH;
    qw[lane] = bf2f(qr[lane]);
    if (DH > 64 && lane + 64 < DH) qw[lane + 64] = bf2f(qr[lane + 64]);
    __builtin_amdgcn_wave_barrier();
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
    for (int d = 0; d < DH; ++d) {
      const float qd = qw[d];
      const bf16_t* kr = KT + (size_t)d * Lp + lane;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < nchunk) acc[c] = fmaf(qd, bf2f(kr[c * 64]), acc[c]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = c * 64 + lane;
      float s = -INFINITY;
      if (c < nchunk && j < L && (!mb || mb[j] != 0) && (!causal || j <= i)) {
        s = rbf(acc[c]);                                 // matmul output, bf16
        if (scale != 1.0f) s = rbf(s * scale);
        if (bias_table) s = rbf(s + bf2f(bias_table[(int64_t)rel_bucket[j - i + L - 1] * heads + h]));
      }
      acc[c] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float e = (c < nchunk && acc[c] > -INFINITY) ? expf(acc[c] - m) : 0.0f;
      acc[c] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nchunk) pw[c * 64 + lane] = rbf(acc[c] * inv);   // softmax(fp32).type_as(bf16)
    __builtin_amdgcn_wave_barrier();
    float o = 0.0f, o2 = 0.0f;
    const bool hi = DH > 64 && lane + 64 < DH;
    for (int j = 0; j < L; ++j) {
      const float pj = pw[j];
      o = fmaf(pj, bf2f(V[(size_t)j * DH + lane]), o);
      if (hi) o2 = fmaf(pj, bf2f(V[(size_t)j * DH + lane + 64]), o2);
    }
    bf16_t* orow = out + ((int64_t)b * L + i) * out_rs + h * DH;
    orow[lane] = f2bf(o);
    if (hi) orow[lane + 64] = f2bf(o2);
    __builtin_amdgcn_wave_barrier();
  }
}

// CLIP text tower activation: x * sigmoid(1.702 x), in place, each op rounded to bf16 as the eager graph does
__global__ __launch_bounds__(256) void quick_gelu_kernel(bf16_t* __restrict__ x, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = bf2f(x[e]);
    const float sg = rbf(1.0f / (1.0f + expf(-rbf(1.702f * v))));
    x[e] = f2bf(v * sg);
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                  bf16_t* __restrict__ o, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    o[e] = f2bf(bf2f(a[e]) * bf2f(b[e]));
}

static unsigned grid_for(int64_t total) {
  const int64_t want = (total + 255) / 256;
  return (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

}  // namespace t5
}  // namespace alg

using namespace alg;

extern "C" int alg_embed_rows(const int64_t* ids, const void* table, void* out, int64_t n, int D, int vocab, void* stream) {
  if (n < 0 || D <= 0 || (D & 7) || vocab <= 0) {
    set_error("alg_embed_rows: bad shape n=%lld D=%d vocab=%d", (long long)n, D, vocab);
    return ALG_EINVAL;
  }
  if (n == 0) return ALG_OK;
  if (!ids || !table || !out) {
    set_error("alg_embed_rows: null pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(t5::embed_kernel, dim3(t5::grid_for(n * (D >> 3))), dim3(256), 0, (hipStream_t)stream, ids,
                     (const bf16_t*)table, (bf16_t*)out, n, D, vocab);
  return check_launch("alg_embed_rows");
}

extern "C" int alg_t5_layernorm(const void* x, const void* weight, void* y, int64_t rows, int D, float eps, void* stream) {
  if (rows < 0 || D <= 0) {
    set_error("alg_t5_layernorm: bad shape");
    return ALG_EINVAL;
  }
  if (rows == 0) return ALG_OK;
  if (!x || !weight || !y) {
    set_error("alg_t5_layernorm: null pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(t5::layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, (bf16_t*)y, rows, D, eps);
  return check_launch("alg_t5_layernorm");
}

extern "C" int alg_attn_bias(const void* q, const void* k, const void* v, void* out, const void* bias_table,
                             const int* rel_bucket, const int* key_mask, int batch, int heads, int head_dim, int L,
                             int64_t qkv_rstride, int64_t out_rstride, float scale, int causal, void* stream) {
  const int Lp = (L + 63) & ~63;
  const size_t lds = (size_t)2 * head_dim * Lp * 2 + 4 * 128 * 4 + (size_t)4 * Lp * 4;
  if (batch < 0 || heads <= 0 || L <= 0 || Lp > 512 || (head_dim != 64 && head_dim != 80) || lds > 160 * 1024 ||
      (qkv_rstride & 7) || (bias_table && !rel_bucket)) {
    set_error("alg_attn_bias: bad shape batch=%d heads=%d head_dim=%d L=%d (head_dim 64 or 80, L <= 512 / 448, strides %% 8 == 0)",
              batch, heads, head_dim, L);
    return ALG_EINVAL;
  }
  if (batch == 0) return ALG_OK;
  if (!q || !k || !v || !out) {
    set_error("alg_attn_bias: null pointer");
    return ALG_EINVAL;
  }
  static PerDeviceOnce attr_set;  // idempotent one-time setup per device; racing first calls both succeed
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
    for (const void* fn : {(const void*)t5::attn_bias_kernel<64>, (const void*)t5::attn_bias_kernel<80>}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        set_error("alg_attn_bias: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return ALG_ELAUNCH;
      }
    }
    device_mark(attr_set, dev_slot);
  }
  const int rows_per_wg = 32;
  const dim3 grid((L + rows_per_wg - 1) / rows_per_wg, batch * heads);
  if (head_dim == 64)
    hipLaunchKernelGGL(t5::attn_bias_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, (const bf16_t*)bias_table, rel_bucket, key_mask,
                       heads, L, Lp, qkv_rstride, out_rstride, rows_per_wg, scale, causal);
  else
    hipLaunchKernelGGL(t5::attn_bias_kernel<80>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, (const bf16_t*)bias_table, rel_bucket, key_mask,
                       heads, L, Lp, qkv_rstride, out_rstride, rows_per_wg, scale, causal);
  return check_launch("alg_attn_bias");
}

extern "C" int alg_quick_gelu(void* x, int64_t numel, void* stream) {
  if (numel < 0) {
    set_error("alg_quick_gelu: bad argument");
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  if (!x) {
    set_error("alg_quick_gelu: null pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(t5::quick_gelu_kernel, dim3(t5::grid_for(numel)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, numel);
  return check_launch("alg_quick_gelu");
}

extern "C" int alg_mul_bf16(const void* a, const void* b, void* out, int64_t numel, void* stream) {
  if (numel < 0) {
    set_error("alg_mul_bf16: bad argument");
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  if (!a || !b || !out) {
    set_error("alg_mul_bf16: null pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(t5::mul_kernel, dim3(t5::grid_for(numel)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)out, numel);
  return check_launch("alg_mul_bf16");
}
