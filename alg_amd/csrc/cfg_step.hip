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

// ---------------------------------------------------------------------------------------------------------------
// Generic loop-body elementwise kernels for the Wan / HunyuanVideo loops, whose schedulers are multi-term linear
// updates (UniPC, flow-match Euler) and whose CFG combine runs in the model dtype.
// ---------------------------------------------------------------------------------------------------------------
namespace alg {

// out = u0 + g * (tx - u) with every intermediate rounded to the prediction dtype (torch eager on bf16 tensors:
// wan:919-924, hy:1254-1261 combine WITHOUT the .float() the CogVideoX loop has)
template <typename T>
__global__ __launch_bounds__(256) void cfg_combine_kernel(const T* __restrict__ pred, T* __restrict__ out, int n_pass,
                                                          int64_t numel, float g) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const T* p_tx = pred + (int64_t)(n_pass - 1) * numel;
  const T* p_u = pred + (int64_t)(n_pass - 2) * numel;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    const float tx = load_as_float<T>(p_tx, e), u = load_as_float<T>(p_u, e);
    const float u0 = n_pass == 3 ? load_as_float<T>(pred, e) : u;
    float d = __fsub_rn(tx, u);
    if (sizeof(T) == 2) d = rbf(d);
    float m = __fmul_rn(g, d);
    if (sizeof(T) == 2) m = rbf(m);
    store_from_float<T>(out, e, __fadd_rn(u0, m));
  }
}

struct LinTerms {
  const void* x[4];
  float c[4];
  int dt[4];
  int n;
};

// out = sum_i c_i * x_i, torch-eager rounding: each product is rounded to its tensor's dtype (scalar * bf16 tensor is
// a bf16 tensor), the running sum is fp32 (term 0 is the fp32 sample in every caller), left to right, unfused
template <typename TO>
__global__ __launch_bounds__(256) void lincomb_kernel(const LinTerms t, TO* __restrict__ out, int64_t numel) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < t.n) {
        const float v = t.dt[i] == ALG_BF16 ? bf2f(((const bf16_t*)t.x[i])[e]) : ((const float*)t.x[i])[e];
        float term = __fmul_rn(t.c[i], v);
        if (t.dt[i] == ALG_BF16) term = rbf(term);
        acc = i == 0 ? term : __fadd_rn(acc, term);
      }
    }
    store_from_float<TO>(out, e, acc);
  }
}

// all-bf16 fast path (the text + image cross-attention sum of the Wan DiT runs over 670 MB): 16-byte accesses
__global__ __launch_bounds__(256) void lincomb_bf16x8_kernel(const LinTerms t, bf16_t* __restrict__ out, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i8 < nvec; i8 += stride) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < t.n) {
        float v[8];
        load8<bf16_t>((const bf16_t*)t.x[i], i8 * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float term = rbf(__fmul_rn(t.c[i], v[k]));
          acc[k] = i == 0 ? term : __fadd_rn(acc[k], term);
        }
      }
    }
    store8<bf16_t>(out, i8 * 8, acc);
  }
}

// UniPC-bh (predict_x0) predictor / corrector update for solver_order <= 2, op order of the published
// multistep_uni_p_bh_update / multistep_uni_c_bh_update (fp32 tensors, fp32 0-dim scalars):
//   x_t_ = r * x - c * m0
//   res  = [has_prev] rho0 * ((m1 - m0) / rk)  (+)  [has_new] rho_new * (m_new - m0)
//   out  = x_t_ - k * res
// `tensor / cpu_scalar` on a GPU is ATen's multiply by the fp32 reciprocal (div_true_cuda), reproduced here.
struct UniPC {
  float r, c, k, rk, rho0, rho_new;
  int has_prev, has_new;
};

__global__ __launch_bounds__(256) void unipc_kernel(const float* __restrict__ x, const float* __restrict__ m0,
                                                    const float* __restrict__ m1, const float* __restrict__ m_new,
                                                    float* __restrict__ out, int64_t numel, UniPC u) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    const float a = m0[e];
    const float xt = __fsub_rn(__fmul_rn(u.r, x[e]), __fmul_rn(u.c, a));
    float res = 0.0f;
    if (u.has_prev) res = __fmul_rn(u.rho0, __fmul_rn(__fsub_rn(m1[e], a), u.rk));
    if (u.has_new) res = __fadd_rn(res, __fmul_rn(u.rho_new, __fsub_rn(m_new[e], a)));
    out[e] = __fsub_rn(xt, __fmul_rn(u.k, res));
  }
}

// CFG batch assembly: out[n, o, a, r] = a < A0 ? src0_n[o, a, r] : src1_n[o, a1_off + a - A0, r], cast to the
// transformer dtype.  One launch replaces the reference's cat([latents]*n) + cat(..., dim) + .to(dtype) chain
// (wan:877-889 channel concat: O=1, A=channels; hy:1146-1160 first-frame token replace: O=channels, A=frames).
struct CatSrc {
  const void* s0[16];
  const void* s1[16];
};

template <typename T0, typename T1, typename TO>
__global__ __launch_bounds__(256) void concat_cast_kernel(const CatSrc src, TO* __restrict__ out, int O, int A0,
                                                          int A1, int64_t R, int64_t s0_ostride, int64_t s1_ostride,
                                                          int a1_off) {
  const int A = A0 + A1;
  const int row = blockIdx.y;  // (n, o, a)
  const int a = row % A, o = (row / A) % O, n = row / (A * O);
  TO* dst = out + (int64_t)row * R;
  if (a < A0) {
    const T0* p = (const T0*)src.s0[n] + (int64_t)o * s0_ostride + (int64_t)a * R;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x)
      store_from_float<TO>(dst, r, load_as_float<T0>(p, r));
  } else {
    const T1* p = (const T1*)src.s1[n] + (int64_t)o * s1_ostride + (int64_t)(a1_off + a - A0) * R;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x)
      store_from_float<TO>(dst, r, load_as_float<T1>(p, r));
  }
}

}  // namespace alg

extern "C" int alg_concat_cast(const void* const* src0, int dtype0, const void* const* src1, int dtype1, int n,
                               int64_t O, int64_t A0, int64_t A1, int64_t R, int64_t s0_ostride, int64_t s1_ostride,
                               int64_t a1_off, void* out, int out_dtype, void* stream) {
  auto okdt = [](int d) { return d == ALG_F32 || d == ALG_BF16; };
  if (!src0 || !src1 || !out || n < 0 || n > 16 || O < 1 || A0 < 0 || A1 < 0 || R < 0 || a1_off < 0 ||
      !okdt(dtype0) || !okdt(dtype1) || !okdt(out_dtype) || n * O * (A0 + A1) > 65535) {
    set_error("alg_concat_cast: bad argument (n=%d O=%lld A0=%lld A1=%lld R=%lld)", n, (long long)O, (long long)A0,
              (long long)A1, (long long)R);
    return ALG_EINVAL;
  }
  if (n == 0 || R == 0 || A0 + A1 == 0) return ALG_OK;
  CatSrc cs;
  for (int i = 0; i < 16; ++i) {
    cs.s0[i] = i < n ? src0[i] : nullptr;
    cs.s1[i] = i < n ? src1[i] : nullptr;
    if (i < n && ((A0 > 0 && !src0[i]) || (A1 > 0 && !src1[i]))) {
      set_error("alg_concat_cast: source %d is null", i);
      return ALG_EINVAL;
    }
  }
  int64_t gx = (R + 1023) / 1024;
  dim3 grid((unsigned)(gx > 64 ? 64 : gx), (unsigned)(n * O * (A0 + A1)));
  hipStream_t s = (hipStream_t)stream;
#define CC(T0, T1, TO)                                                                                             \
  hipLaunchKernelGGL((concat_cast_kernel<T0, T1, TO>), grid, dim3(256), 0, s, cs, (TO*)out, (int)O, (int)A0, (int)A1, \
                     R, s0_ostride, s1_ostride, (int)a1_off)
#define CC1(T0, T1)                                                                                                \
  do {                                                                                                             \
    if (out_dtype == ALG_F32) CC(T0, T1, float);                                                                   \
    else CC(T0, T1, bf16_t);                                                                                       \
  } while (0)
  if (dtype0 == ALG_F32 && dtype1 == ALG_F32) CC1(float, float);
  else if (dtype0 == ALG_F32) CC1(float, bf16_t);
  else if (dtype1 == ALG_F32) CC1(bf16_t, float);
  else CC1(bf16_t, bf16_t);
#undef CC1
#undef CC
  return check_launch("alg_concat_cast");
}

extern "C" int alg_unipc_update(const float* x, const float* m0, const float* m1, const float* m_new, float* out,
                                int64_t numel, float r, float c, float k, float rk, float rho0, float rho_new,
                                void* stream) {
  if (!x || !m0 || !out || numel < 0) {
    set_error("alg_unipc_update: bad argument (numel=%lld)", (long long)numel);
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  UniPC u{r, c, k, 1.0f / rk, rho0, rho_new, m1 != nullptr, m_new != nullptr};  // u.rk holds the reciprocal
  int64_t want = (numel + 255) / 256;
  const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
  hipLaunchKernelGGL(unipc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, m0, m1, m_new, out, numel, u);
  return check_launch("alg_unipc_update");
}

extern "C" int alg_cfg_combine(const void* pred, void* out, int dtype, int n_pass, int64_t numel, float guidance_scale,
                               void* stream) {
  if (numel < 0 || n_pass < 2 || n_pass > 3 || (dtype != ALG_F32 && dtype != ALG_BF16)) {
    set_error("alg_cfg_combine: bad argument (n_pass=%d numel=%lld dtype=%d)", n_pass, (long long)numel, dtype);
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;  // empty tensors carry null data pointers
  if (!pred || !out) {
    set_error("alg_cfg_combine: null pointer");
    return ALG_EINVAL;
  }
  int64_t want = (numel + 255) / 256;
  const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
  if (dtype == ALG_F32)
    hipLaunchKernelGGL(cfg_combine_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)pred,
                       (float*)out, n_pass, numel, guidance_scale);
  else
    hipLaunchKernelGGL(cfg_combine_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred,
                       (bf16_t*)out, n_pass, numel, guidance_scale);
  return check_launch("alg_cfg_combine");
}

extern "C" int alg_lincomb(const void* const* xs, const float* coefs, const int* dtypes, int n_terms, void* out,
                           int out_dtype, int64_t numel, void* stream) {
  if (!xs || !coefs || !dtypes || !out || n_terms < 1 || n_terms > 4 || numel < 0 ||
      (out_dtype != ALG_F32 && out_dtype != ALG_BF16)) {
    set_error("alg_lincomb: bad argument (n_terms=%d numel=%lld)", n_terms, (long long)numel);
    return ALG_EINVAL;
  }
  LinTerms t;
  t.n = n_terms;
  for (int i = 0; i < 4; ++i) {
    t.x[i] = i < n_terms ? xs[i] : nullptr;
    t.c[i] = i < n_terms ? coefs[i] : 0.0f;
    t.dt[i] = i < n_terms ? dtypes[i] : ALG_F32;
    if (i < n_terms && (!xs[i] || (dtypes[i] != ALG_F32 && dtypes[i] != ALG_BF16))) {
      set_error("alg_lincomb: term %d is null or has an unsupported dtype", i);
      return ALG_EINVAL;
    }
  }
  if (numel == 0) return ALG_OK;
  bool all_bf16 = out_dtype == ALG_BF16 && numel % 8 == 0 && ((uintptr_t)out & 15) == 0;
  for (int i = 0; i < n_terms; ++i) all_bf16 = all_bf16 && dtypes[i] == ALG_BF16 && ((uintptr_t)xs[i] & 15) == 0;
  if (all_bf16) {
    const int64_t nvec = numel / 8, wantv = (nvec + 255) / 256;
    hipLaunchKernelGGL(lincomb_bf16x8_kernel, dim3((unsigned)(wantv > 8192 ? 8192 : wantv)), dim3(256), 0,
                       (hipStream_t)stream, t, (bf16_t*)out, nvec);
    return check_launch("alg_lincomb");
  }
  int64_t want = (numel + 255) / 256;
  const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
  if (out_dtype == ALG_F32)
    hipLaunchKernelGGL(lincomb_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, t, (float*)out, numel);
  else
    hipLaunchKernelGGL(lincomb_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, t, (bf16_t*)out, numel);
  return check_launch("alg_lincomb");
}
