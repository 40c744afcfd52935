// Input/output plumbing kernels of the DiT forward: patch gather (with the CFG batch assembly of
// cog:1060-1070 folded in -- no materialised torch.cat), unpatchify and the sinusoidal timestep projection.
#include "common.h"

namespace alg {

constexpr int MAX_SAMPLES = 8;
struct CondPtrs {
  const bf16_t* p[MAX_SAMPLES];
};

// thread = (n, f, c2, y, gx): reads p consecutive x from latents (c2 < C) or cond[n] (c2 >= C), writes p
// consecutive k entries of row (f / pt, y / p, gx).  pt = 1: CogVideoX 1.0 (Conv2d patch embed, k = (c, py, px));
// pt > 1: CogVideoX 1.5 (Linear over (c, t, py, px), frames folded in groups of pt).
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ lat, int64_t lat_bs,
                                                       const CondPtrs cond, bf16_t* __restrict__ out, int n_samples,
                                                       int F, int C, int H, int W, int p, int pt) {
  const int gw = W / p, gh = H / p;
  const int64_t per_sample = (int64_t)F * 2 * C * H * gw;
  const int64_t total = per_sample * n_samples;
  const int K = 2 * C * pt * p * p;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int gx = (int)(r % gw); r /= gw;
    const int y = (int)(r % H); r /= H;
    const int c2 = (int)(r % (2 * C)); r /= (2 * C);
    const int f = (int)(r % F);
    const int n = (int)(r / F);
    const bf16_t* src = c2 < C ? lat + (int64_t)n * lat_bs + (((int64_t)f * C + c2) * H + y) * W
                               : cond.p[n] + (((int64_t)f * C + (c2 - C)) * H + y) * W;
    const int64_t tok = ((int64_t)(f / pt) * gh + y / p) * gw + gx;
    bf16_t* dst = out + ((int64_t)n * (F / pt) * gh * gw + tok) * K + ((c2 * pt + f % pt) * p + (y % p)) * p;
    for (int px = 0; px < p; ++px) dst[px] = src[gx * p + px];
  }
}

// thread = output element (n, f, c, y, x)
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                         int n_samples, int F, int C, int H, int W, int p, int pt) {
  const int gw = W / p, gh = H / p;
  const int K = C * pt * p * p;
  const int64_t total = (int64_t)n_samples * F * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C); r /= C;
    const int f = (int)(r % F);
    const int n = (int)(r / F);
    const int64_t tok = ((int64_t)(f / pt) * gh + y / p) * gw + x / p;
    out[i] = in[((int64_t)n * (F / pt) * gh * gw + tok) * K + ((c * pt + f % pt) * p + (y % p)) * p + (x % p)];
  }
}

// diffusers get_timestep_embedding(scale=1, max_period=10000, downscale_freq_shift=0), unfused fp32 like eager
__global__ __launch_bounds__(256) void timestep_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int n,
                                                       int dim, int flip) {
  const int half = dim / 2;
  const int total = n * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int s = i / half, k = i - s * half;
    const float e = __fdiv_rn(__fmul_rn(-9.210340371976184f, (float)k), (float)half);
    const float a = __fmul_rn(t[s], expf(e));
    const float sv = sinf(a), cv = cosf(a);
    bf16_t* o = out + (int64_t)s * dim;
    o[k] = f2bf(flip ? cv : sv);
    o[half + k] = f2bf(flip ? sv : cv);
  }
}

}  // namespace alg

using namespace alg;

extern "C" int alg_patchify_t(const void* latents, int64_t lat_bstride, const void* const* cond_ptrs, void* out,
                              int n_samples, int frames, int C, int H, int W, int p, int p_t, void* stream);

extern "C" int alg_patchify(const void* latents, int64_t lat_bstride, const void* const* cond_ptrs, void* out,
                            int n_samples, int frames, int C, int H, int W, int p, void* stream) {
  return alg_patchify_t(latents, lat_bstride, cond_ptrs, out, n_samples, frames, C, H, W, p, 1, stream);
}

extern "C" int alg_patchify_t(const void* latents, int64_t lat_bstride, const void* const* cond_ptrs, void* out,
                              int n_samples, int frames, int C, int H, int W, int p, int p_t, void* stream) {
  if (!latents || !cond_ptrs || !out || n_samples <= 0 || n_samples > MAX_SAMPLES || frames <= 0 || C <= 0 || H <= 0 ||
      W <= 0 || p <= 0 || H % p || W % p || p_t <= 0 || frames % p_t) {
    set_error("alg_patchify: bad argument (n=%d F=%d C=%d H=%d W=%d p=%d; n <= %d, H,W %% p == 0)", n_samples, frames,
              C, H, W, p, MAX_SAMPLES);
    return ALG_EINVAL;
  }
  CondPtrs cp;
  for (int i = 0; i < MAX_SAMPLES; ++i) cp.p[i] = i < n_samples ? (const bf16_t*)cond_ptrs[i] : nullptr;
  for (int i = 0; i < n_samples; ++i)
    if (!cp.p[i]) {
      set_error("alg_patchify: cond_ptrs[%d] is null", i);
      return ALG_EINVAL;
    }
  const int64_t total = (int64_t)n_samples * frames * 2 * C * H * (W / p);
  int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
  hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)latents,
                     lat_bstride, cp, (bf16_t*)out, n_samples, frames, C, H, W, p, p_t);
  return check_launch("alg_patchify");
}

extern "C" int alg_unpatchify_t(const void* in, void* out, int n_samples, int frames, int C, int H, int W, int p, int p_t,
                                void* stream);

extern "C" int alg_unpatchify(const void* in, void* out, int n_samples, int frames, int C, int H, int W, int p,
                              void* stream) {
  return alg_unpatchify_t(in, out, n_samples, frames, C, H, W, p, 1, stream);
}

extern "C" int alg_unpatchify_t(const void* in, void* out, int n_samples, int frames, int C, int H, int W, int p, int p_t,
                                void* stream) {
  if (!in || !out || n_samples <= 0 || frames <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p || p_t <= 0 ||
      frames % p_t) {
    set_error("alg_unpatchify: bad argument (n=%d F=%d C=%d H=%d W=%d p=%d)", n_samples, frames, C, H, W, p);
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)n_samples * frames * C * H * W;
  int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
  hipLaunchKernelGGL(unpatchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     (bf16_t*)out, n_samples, frames, C, H, W, p, p_t);
  return check_launch("alg_unpatchify");
}

extern "C" int alg_timestep_embedding(const float* t, void* out, int n, int dim, int flip_sin_to_cos, void* stream) {
  if (!t || !out || n <= 0 || dim <= 0 || dim % 2) {
    set_error("alg_timestep_embedding: bad argument (n=%d dim=%d)", n, dim);
    return ALG_EINVAL;
  }
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(timestep_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (bf16_t*)out, n,
                     dim, flip_sin_to_cos ? 1 : 0);
  return check_launch("alg_timestep_embedding");
}
