// Low-pass filters for planes that do not fit the LDS-resident kernels of lowpass.hip (the pixel-space ALG branch,
// cog:628-680 / wan:493-540 / hy:703-768: the RGB image [B, 3, H, W] is filtered, then re-encoded by the VAE).  Same
// arithmetic, operation for operation (strict fp32 tap weights, acc = s0*w0 then fma over ascending taps, fp32
// intermediates, optional bf16 rounding between the two interpolate calls), as separable passes through a stream-ordered
// global workspace; tap weights are recomputed per output instead of tabulated.
#include "common.h"

namespace alg {
namespace big {

struct Tap {
  int lo, n;
  float center, invscale, tot;
};

// lowpass.hip build_taps, one output at a time
__device__ __forceinline__ float tap_raw(const Tap& t, int j) {
  float x = __fmul_rn(__fadd_rn(__fsub_rn((float)(j + t.lo), t.center), 0.5f), t.invscale);
  x = fabsf(x);
  return x < 1.0f ? __fsub_rn(1.0f, x) : 0.0f;
}

__device__ __forceinline__ Tap tap_of(int i, int in_size, int out_size) {
  const float scale = __fdiv_rn((float)in_size, (float)out_size);
  const float support = scale >= 1.0f ? scale : 1.0f;
  Tap t;
  t.invscale = scale >= 1.0f ? __fdiv_rn(1.0f, scale) : 1.0f;
  t.center = __fmul_rn(scale, (float)i + 0.5f);
  int lo = (int)__fadd_rn(__fsub_rn(t.center, support), 0.5f);
  lo = lo > 0 ? lo : 0;
  int hi = (int)__fadd_rn(__fadd_rn(t.center, support), 0.5f);
  hi = hi < in_size ? hi : in_size;
  const int maxt = (int)ceilf(support) * 2 + 1;
  int n = hi - lo;
  n = n < 0 ? 0 : (n > maxt ? maxt : n);
  t.lo = lo, t.n = n;
  float tot = 0.0f;
  for (int j = 0; j < n; ++j) tot = __fadd_rn(tot, tap_raw(t, j));
  t.tot = tot;
  return t;
}

__device__ __forceinline__ float tap_w(const Tap& t, int j) {
  const float w = tap_raw(t, j);
  return t.tot != 0.0f ? __fdiv_rn(w, t.tot) : w;
}

// dst[p][r][o] = sum_j w[o][j] * src[p][r][lo(o) + j]      (contiguous axis)
template <typename TI>
__global__ __launch_bounds__(256) void rows_kernel(const TI* __restrict__ src, float* __restrict__ dst, int rows, int n_in,
                                                   int n_out, int rows_per_block) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  const int64_t p = blockIdx.z;
  const Tap t = tap_of(o, n_in, n_out);
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  for (int r = r0; r < r1; ++r) {
    const TI* s = src + (p * rows + r) * n_in + t.lo;
    float acc = t.n > 0 ? load_as_float<TI>(s, 0) * tap_w(t, 0) : 0.0f;
    for (int j = 1; j < t.n; ++j) acc = fmaf(load_as_float<TI>(s, j), tap_w(t, j), acc);
    dst[(p * rows + r) * n_out + o] = acc;
  }
}

// dst[p][o][c] = sum_j w[o][j] * src[p][lo(o) + j][c]      (strided axis)
template <typename TO, bool ROUND>
__global__ __launch_bounds__(256) void cols_kernel(const float* __restrict__ src, TO* __restrict__ dst, int n_in, int n_out,
                                                   int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int o = blockIdx.y;
  const int64_t p = blockIdx.z;
  const Tap t = tap_of(o, n_in, n_out);
  const float* s = src + (p * n_in + t.lo) * cols + c;
  float acc = t.n > 0 ? s[0] * tap_w(t, 0) : 0.0f;
  for (int j = 1; j < t.n; ++j) acc = fmaf(s[(int64_t)j * cols], tap_w(t, j), acc);
  if (ROUND) acc = rbf(acc);
  store_from_float<TO>(dst, (p * n_out + o) * cols + c, acc);
}

__device__ __forceinline__ int reflect(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// g[j] = exp(-0.5 (x / sigma)^2) / sum, as lowpass.hip's gaussian_kernel builds it
__global__ void gauss_weights_kernel(float* g, int ksize, float sigma) {   // one workgroup, any ksize
  const int tid = threadIdx.x;
  for (int j = tid; j < ksize; j += blockDim.x) {
    const float x = (float)j - 0.5f * (float)(ksize - 1);
    const float q = __fdiv_rn(x, sigma);
    g[j] = expf(__fmul_rn(-0.5f, __fmul_rn(q, q)));
  }
  __syncthreads();
  float tot = 0.0f;
  for (int j = 0; j < ksize; ++j) tot = __fadd_rn(tot, g[j]);   // the sequential sum, in every thread
  __syncthreads();
  for (int j = tid; j < ksize; j += blockDim.x) g[j] = __fdiv_rn(g[j], tot);
}

template <typename TI>
__global__ __launch_bounds__(256) void gauss_rows_kernel(const TI* __restrict__ src, float* __restrict__ dst,
                                                         const float* __restrict__ g, int H, int W, int ksize) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y = blockIdx.y, pad = ksize / 2;
  const int64_t p = blockIdx.z;
  const TI* row = src + (p * H + y) * W;
  float acc = 0.0f;
  for (int j = 0; j < ksize; ++j) acc = fmaf(g[j], load_as_float<TI>(row, reflect(x - pad + j, W)), acc);
  dst[(p * H + y) * W + x] = acc;
}

template <typename TO>
__global__ __launch_bounds__(256) void gauss_cols_kernel(const float* __restrict__ src, TO* __restrict__ dst,
                                                         const float* __restrict__ g, int H, int W, int ksize) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y = blockIdx.y, pad = ksize / 2;
  const int64_t p = blockIdx.z;
  const float* pl = src + p * H * W;
  float acc = 0.0f;
  for (int i = 0; i < ksize; ++i) acc = fmaf(g[i], pl[(int64_t)reflect(y - pad + i, H) * W + x], acc);
  store_from_float<TO>(dst, (p * H + y) * W + x, acc);
}

// The intermediates of the global-memory passes live in a workspace the CALLER provides (alg_*_workspace_bytes): the
// library never allocates.
static int ws_check(float** p, size_t floats, void* workspace, int64_t workspace_bytes, const char* who) {
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < (int64_t)(floats * sizeof(float))) {
    set_error("%s: this shape runs through global-memory passes and needs a 16-byte aligned workspace of %zu bytes "
              "(%s_workspace_bytes); got %p / %lld bytes", who, floats * sizeof(float), who, workspace,
              (long long)workspace_bytes);
    return ALG_EINVAL;
  }
  *p = (float*)workspace;
  return ALG_OK;
}

}  // namespace big

template <typename T>
static int down_up_big_t(const T* in, T* out, int64_t planes, int H, int W, int h1, int w1, int round_mid, void* workspace,
                         int64_t workspace_bytes, hipStream_t s) {
  if (planes > 65535 || H > 65535 || h1 > 65535) {
    set_error("alg_down_up: %lld planes of %dx%d exceed the global-memory path's grid", (long long)planes, H, W);
    return ALG_ELIMIT;
  }
  float* ws = nullptr;
  const size_t n1 = (size_t)planes * H * w1, n2 = (size_t)planes * h1 * w1, n3 = (size_t)planes * h1 * W;
  if (int rc = big::ws_check(&ws, n1 + n2 + n3, workspace, workspace_bytes, "alg_down_up")) return rc;
  float *T1 = ws, *T2 = ws + n1, *T3 = T2 + n2;
  const int rpb = 8;
  // first interpolate call (lp:53): W pass, H pass (result optionally rounded to bf16 like the tensor in between)
  hipLaunchKernelGGL(big::rows_kernel<T>, dim3((w1 + 255) / 256, (H + rpb - 1) / rpb, (unsigned)planes), dim3(256), 0, s,
                     in, T1, H, W, w1, rpb);
  if (round_mid)
    hipLaunchKernelGGL((big::cols_kernel<float, true>), dim3((w1 + 255) / 256, h1, (unsigned)planes), dim3(256), 0, s,
                       (const float*)T1, T2, H, h1, w1);
  else
    hipLaunchKernelGGL((big::cols_kernel<float, false>), dim3((w1 + 255) / 256, h1, (unsigned)planes), dim3(256), 0, s,
                       (const float*)T1, T2, H, h1, w1);
  // second interpolate call (lp:54)
  hipLaunchKernelGGL(big::rows_kernel<float>, dim3((W + 255) / 256, (h1 + rpb - 1) / rpb, (unsigned)planes), dim3(256), 0,
                     s, (const float*)T2, T3, h1, w1, W, rpb);
  hipLaunchKernelGGL((big::cols_kernel<T, false>), dim3((W + 255) / 256, H, (unsigned)planes), dim3(256), 0, s,
                     (const float*)T3, out, h1, H, W);
  return check_launch("alg_down_up");
}

int64_t down_up_big_bytes(int64_t planes, int H, int W, int h1, int w1) {
  return (int64_t)sizeof(float) * planes * ((int64_t)H * w1 + (int64_t)h1 * w1 + (int64_t)h1 * W);
}

int64_t gaussian_big_bytes(int64_t planes, int H, int W, int ksize) {
  return (int64_t)sizeof(float) * (planes * H * W + ((ksize + 255) & ~255));
}

int down_up_big(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
                void* workspace, int64_t workspace_bytes, hipStream_t s) {
  if (dtype == ALG_F32)
    return down_up_big_t<float>((const float*)in, (float*)out, planes, H, W, h1, w1, 0, workspace, workspace_bytes, s);
  return down_up_big_t<bf16_t>((const bf16_t*)in, (bf16_t*)out, planes, H, W, h1, w1, round_mid, workspace, workspace_bytes, s);
}

template <typename T>
static int gaussian_big_t(const T* in, T* out, int64_t planes, int H, int W, int ksize, float sigma, void* workspace,
                          int64_t workspace_bytes, hipStream_t s) {
  if (planes > 65535 || H > 65535) {
    set_error("alg_gaussian_blur: %lld planes of %dx%d exceed the global-memory path's grid", (long long)planes, H, W);
    return ALG_ELIMIT;
  }
  float* ws = nullptr;
  const size_t n = (size_t)planes * H * W;
  if (int rc = big::ws_check(&ws, n + (size_t)((ksize + 255) & ~255), workspace, workspace_bytes, "alg_gaussian_blur")) return rc;
  float* g = ws + n;
  hipLaunchKernelGGL(big::gauss_weights_kernel, dim3(1), dim3(256), 0, s, g, ksize, sigma);
  const dim3 grid((W + 255) / 256, H, (unsigned)planes);
  hipLaunchKernelGGL(big::gauss_rows_kernel<T>, grid, dim3(256), 0, s, in, ws, (const float*)g, H, W, ksize);
  hipLaunchKernelGGL(big::gauss_cols_kernel<T>, grid, dim3(256), 0, s, (const float*)ws, out, (const float*)g, H, W, ksize);
  return check_launch("alg_gaussian_blur");
}

int gaussian_big(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype,
                 void* workspace, int64_t workspace_bytes, hipStream_t s) {
  if (dtype == ALG_F32)
    return gaussian_big_t<float>((const float*)in, (float*)out, planes, H, W, ksize, sigma, workspace, workspace_bytes, s);
  return gaussian_big_t<bf16_t>((const bf16_t*)in, (bf16_t*)out, planes, H, W, ksize, sigma, workspace, workspace_bytes, s);
}

}  // namespace alg
