// Fused CFG combine + CogVideoXDDIMScheduler.step (v-prediction, eta = 0) + cast, in place on latents.
// Replaces pipeline_cogvideox_image2video_lowpass.py:1091-1123 (float(); chunk; combine; step; .to(dtype)).
// HBM-bound: (n_pass * sizeof(pred) + 2 * sizeof(lat)) bytes per element, 16-byte accesses per lane.
//
// Rounding points follow what torch eager does on the reference's tensors: the combine runs in fp32 as
// three separate ops (sub, mul by python scalar, add -> no FMA contraction); in the step the 0-dim fp64
// scheduler scalars do not promote a dimensioned tensor, so `sqrt_alpha_t * sample` and `coef_a * sample`
// are rounded to the latents' dtype (bf16) before meeting the fp32 terms.
#include "common.h"

namespace alg {

struct StepCoef {
  float g, sa, sb, ca, cb;
  int n_pass;
};

template <bool LAT_BF16>
__device__ __forceinline__ float step_one(const StepCoef c, float u0, float u, float tx, float x) {
  float v;
  if (c.n_pass == 1) {
    v = tx;
  } else {
    v = __fadd_rn(c.n_pass == 3 ? u0 : u, __fmul_rn(c.g, __fsub_rn(tx, u)));
  }
  float t1 = __fmul_rn(c.sa, x);
  float t2 = __fmul_rn(c.ca, x);
  if (LAT_BF16) {
    t1 = rbf(t1);
    t2 = rbf(t2);
  }
  const float x0 = __fsub_rn(t1, __fmul_rn(c.sb, v));
  return __fadd_rn(t2, __fmul_rn(c.cb, x0));
}

// 8 consecutive elements as floats
template <typename T>
__device__ __forceinline__ void load8(const T* p, int64_t e, float (&f)[8]) {
  if constexpr (sizeof(T) == 4) {
    const float4 a = *(const float4*)(p + e), b = *(const float4*)(p + e + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const uint4 v = *(const uint4*)(p + e);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(u[k] << 16);
      f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
    }
  }
}

template <typename T>
__device__ __forceinline__ void store8(T* p, int64_t e, const float (&f)[8]) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)(p + e) = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + e + 4) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    *(uint4*)(p + e) = v;
  }
}

template <typename TP, typename TL, bool VEC>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const TP* __restrict__ pred, TL* __restrict__ lat,
                                                       int64_t numel, StepCoef c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // "text" is always the last chunk, "uncond" the one before it, "uncond_init" the first of three
  const TP* p_tx = pred + (int64_t)(c.n_pass - 1) * numel;
  const TP* p_u = c.n_pass >= 2 ? pred + (int64_t)(c.n_pass - 2) * numel : pred;
  const TP* p_u0 = pred;
  constexpr bool LB = sizeof(TL) == 2;
  if (VEC) {
    const int64_t nvec = numel / 8;
    for (int64_t i = gid; i < nvec; i += stride) {
      float tx[8], u[8], u0[8], x[8], r[8];
      load8<TP>(p_tx, i * 8, tx);
      if (c.n_pass >= 2) load8<TP>(p_u, i * 8, u);
      if (c.n_pass == 3) load8<TP>(p_u0, i * 8, u0);
      load8<TL>(lat, i * 8, x);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        r[k] = step_one<LB>(c, c.n_pass == 3 ? u0[k] : 0.0f, c.n_pass >= 2 ? u[k] : 0.0f, tx[k], x[k]);
      store8<TL>(lat, i * 8, r);
    }
  } else {
    for (int64_t e = gid; e < numel; e += stride) {
      const float tx = load_as_float<TP>(p_tx, e);
      const float u = c.n_pass >= 2 ? load_as_float<TP>(p_u, e) : 0.0f;
      const float u0 = c.n_pass == 3 ? load_as_float<TP>(p_u0, e) : 0.0f;
      store_from_float<TL>(lat, e, step_one<LB>(c, u0, u, tx, load_as_float<TL>(lat, e)));
    }
  }
}

}  // namespace alg

using namespace alg;

extern "C" int alg_cfg_ddim_step(const void* pred, int pred_dtype, void* latents, int lat_dtype, int n_pass,
                                 int64_t numel, float guidance_scale, float sqrt_alpha_t, float sqrt_beta_t,
                                 float coef_a, float coef_b, void* stream) {
  if (!pred || !latents || numel < 0 || n_pass < 1 || n_pass > 3) {
    set_error("alg_cfg_ddim_step: bad argument (n_pass=%d numel=%lld)", n_pass, (long long)numel);
    return ALG_EINVAL;
  }
  if ((pred_dtype != ALG_F32 && pred_dtype != ALG_BF16) || (lat_dtype != ALG_F32 && lat_dtype != ALG_BF16)) {
    set_error("alg_cfg_ddim_step: unsupported dtype codes %d/%d", pred_dtype, lat_dtype);
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t psz = pred_dtype == ALG_F32 ? 4 : 2;
  // vector path: every chunk base 16-byte aligned and whole 8-element groups
  const bool vec = (numel % 8 == 0) && (((uintptr_t)pred) % 16 == 0) && (((uintptr_t)latents) % 16 == 0) &&
                   ((numel * psz) % 16 == 0);
  const int64_t work = vec ? numel / 8 : numel;
  int64_t want = (work + 255) / 256;
  unsigned grid = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  StepCoef c{guidance_scale, sqrt_alpha_t, sqrt_beta_t, coef_a, coef_b, n_pass};
#define LAUNCH(TP, TL)                                                                                              \
  do {                                                                                                              \
    if (vec)                                                                                                        \
      hipLaunchKernelGGL((cfg_ddim_kernel<TP, TL, true>), dim3(grid), dim3(256), 0, s, (const TP*)pred, (TL*)latents, \
                         numel, c);                                                                                 \
    else                                                                                                            \
      hipLaunchKernelGGL((cfg_ddim_kernel<TP, TL, false>), dim3(grid), dim3(256), 0, s, (const TP*)pred,             \
                         (TL*)latents, numel, c);                                                                   \
  } while (0)
  if (pred_dtype == ALG_F32 && lat_dtype == ALG_F32) LAUNCH(float, float);
  else if (pred_dtype == ALG_F32) LAUNCH(float, bf16_t);
  else if (lat_dtype == ALG_F32) LAUNCH(bf16_t, float);
  else LAUNCH(bf16_t, bf16_t);
#undef LAUNCH
  return check_launch("alg_cfg_ddim_step");
}
