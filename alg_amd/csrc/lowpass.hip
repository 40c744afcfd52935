// ALG low-pass filters for gfx950: antialiased-bilinear down/up (lp_utils.py:49-54) and separable
// gaussian blur (lp_utils.py:40-47).  One workgroup owns one (H, W) plane: the plane is read from HBM
// once (16-byte loads when aligned), staged in LDS as fp32, every separable pass runs LDS->LDS, and the
// result is written to HBM once -- algorithmic traffic = 2 * planes * H * W * sizeof(dtype).
//
// The antialias tap tables are computed in-kernel in strict (unfused) fp32, operation for operation what
// ATen's _upsample_bilinear2d_aa does for float/bf16 tensors (support = max(scale, 1), triangle filter,
// per-output normalisation), so tap sets match ATen's and results agree to fp32 rounding.
#include <stdlib.h>

#include "common.h"

namespace alg {

struct TapTable {
  int* xmin;    // [n_out]
  int* xsize;   // [n_out]
  float* w;     // [n_out][taps]
  int taps;
};

__host__ __device__ inline int aa_max_taps(int in_size, int out_size) {
  float scale = (float)in_size / (float)out_size;
  float support = scale >= 1.0f ? scale : 1.0f;
  return (int)ceilf(support) * 2 + 1;
}

// strict fp32, no FMA contraction: __f*_rn intrinsics are never fused
__device__ void build_taps(const TapTable t, int in_size, int out_size, int tid, int nthreads) {
  const float scale = __fdiv_rn((float)in_size, (float)out_size);
  const float support = scale >= 1.0f ? scale : 1.0f;
  const float invscale = scale >= 1.0f ? __fdiv_rn(1.0f, scale) : 1.0f;
  for (int i = tid; i < out_size; i += nthreads) {
    const float center = __fmul_rn(scale, (float)i + 0.5f);
    int lo = (int)__fadd_rn(__fsub_rn(center, support), 0.5f);
    lo = lo > 0 ? lo : 0;
    int hi = (int)__fadd_rn(__fadd_rn(center, support), 0.5f);
    hi = hi < in_size ? hi : in_size;
    int n = hi - lo;
    n = n < 0 ? 0 : (n > t.taps ? t.taps : n);
    float* w = t.w + (size_t)i * t.taps;
    float tot = 0.0f;
    for (int j = 0; j < n; ++j) {
      float x = __fmul_rn(__fadd_rn(__fsub_rn((float)(j + lo), center), 0.5f), invscale);
      x = fabsf(x);
      float wj = x < 1.0f ? __fsub_rn(1.0f, x) : 0.0f;
      w[j] = wj;
      tot = __fadd_rn(tot, wj);
    }
    for (int j = 0; j < n; ++j) w[j] = tot != 0.0f ? __fdiv_rn(w[j], tot) : w[j];
    for (int j = n; j < t.taps; ++j) w[j] = 0.0f;
    t.xmin[i] = lo;
    t.xsize[i] = n;
  }
}

__device__ __forceinline__ TapTable carve_taps(char*& p, int n_out, int taps) {
  TapTable t;
  t.taps = taps;
  t.xmin = (int*)p;
  p += sizeof(int) * n_out;
  t.xsize = (int*)p;
  p += sizeof(int) * n_out;
  t.w = (float*)p;
  p += sizeof(float) * (size_t)n_out * taps;
  return t;
}

// Both 1-D passes give each thread ONE output column (row passes) or ONE plane column (column passes) and let it walk
// the other axis, so the per-output integer divisions are gone and the tap weights of a row pass sit in registers.
// The accumulation order is the reference's: acc = s0*w0, then fma over the taps in ascending order.
constexpr int REG_TAPS = 12;

// dst[r][o] = sum_j w[o][j] * src[r][xmin[o] + j]      (resize along the contiguous axis)
__device__ __forceinline__ void pass_rows(const float* src, int src_ld, float* dst, int dst_ld, int rows, int n_out,
                                          const TapTable t, int tid, int nthreads) {
  const bool fits = n_out <= nthreads;
  const int rstep = fits ? nthreads / n_out : 1;
  const int r0 = fits ? tid / n_out : 0;
  for (int o = fits ? tid - r0 * n_out : tid; o < n_out; o += nthreads) {
    if (r0 >= rstep) break;  // leftover threads of the last partial row group
    const int n = t.xsize[o];
    const float* wp = t.w + (size_t)o * t.taps;
    const float* s0 = src + t.xmin[o];
    if (t.taps <= REG_TAPS) {
      float w[REG_TAPS];
#pragma unroll
      for (int j = 0; j < REG_TAPS; ++j) w[j] = j < t.taps ? wp[j] : 0.0f;
      for (int r = r0; r < rows; r += rstep) {
        const float* s = s0 + (size_t)r * src_ld;
        float acc = n > 0 ? s[0] * w[0] : 0.0f;
#pragma unroll
        for (int j = 1; j < REG_TAPS; ++j)
          if (j < n) acc = fmaf(s[j], w[j], acc);
        dst[(size_t)r * dst_ld + o] = acc;
      }
    } else {
      for (int r = r0; r < rows; r += rstep) {
        const float* s = s0 + (size_t)r * src_ld;
        float acc = n > 0 ? s[0] * wp[0] : 0.0f;
        for (int j = 1; j < n; ++j) acc = fmaf(s[j], wp[j], acc);
        dst[(size_t)r * dst_ld + o] = acc;
      }
    }
  }
}

// out(o, c) = sum_j w[o][j] * src[xmin[o] + j][c]      (resize along the strided axis); `store(o, c, value)`
template <typename Store>
__device__ __forceinline__ void pass_cols(const float* src, int src_ld, int cols, int n_out, const TapTable t, int tid,
                                          int nthreads, Store store) {
  const bool fits = cols <= nthreads;
  const int ostep = fits ? nthreads / cols : 1;
  const int o0 = fits ? tid / cols : 0;
  for (int c = fits ? tid - o0 * cols : tid; c < cols; c += nthreads) {
    if (o0 >= ostep) break;
    for (int o = o0; o < n_out; o += ostep) {
      const int n = t.xsize[o];
      const float* s = src + (size_t)t.xmin[o] * src_ld + c;
      const float* w = t.w + (size_t)o * t.taps;
      float acc = n > 0 ? s[0] * w[0] : 0.0f;
      for (int j = 1; j < n; ++j) acc = fmaf(s[(size_t)j * src_ld], w[j], acc);
      store(o, c, acc);
    }
  }
}

template <typename T>
__device__ __forceinline__ void load_plane(const T* g, float* lds, int n, int tid, int nthreads) {
  constexpr int V = 16 / sizeof(T);
  if ((((uintptr_t)g) & 15) == 0) {
    const int nv = n / V;
    const uint4* gv = (const uint4*)g;
    for (int i = tid; i < nv; i += nthreads) {
      uint4 v = gv[i];
      if constexpr (sizeof(T) == 4) {
        float4 f = *(float4*)&v;
        *(float4*)(lds + (size_t)i * 4) = f;
      } else {
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          lds[(size_t)i * 8 + 2 * k] = __uint_as_float(u[k] << 16);
          lds[(size_t)i * 8 + 2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
        }
      }
    }
    for (int i = nv * V + tid; i < n; i += nthreads) lds[i] = load_as_float<T>(g, i);
  } else {
    for (int i = tid; i < n; i += nthreads) lds[i] = load_as_float<T>(g, i);
  }
}

// ---------------------------------------------------------------------------------------------
// down_up
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void down_up_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                      int h1, int w1, int round_mid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t plane = blockIdx.x;
  // LDS carve: X/T3 [H*W] | T1 [H*w1] | T2 [h1*w1] | tap tables
  float* X = (float*)smem;
  float* T1 = X + (size_t)H * W;
  float* T2 = T1 + (size_t)H * w1;
  char* p = (char*)(T2 + (size_t)h1 * w1);
  TapTable dw = carve_taps(p, w1, aa_max_taps(W, w1));   // W  -> w1
  TapTable dh = carve_taps(p, h1, aa_max_taps(H, h1));   // H  -> h1
  TapTable uw = carve_taps(p, W, aa_max_taps(w1, W));    // w1 -> W
  TapTable uh = carve_taps(p, H, aa_max_taps(h1, H));    // h1 -> H

  build_taps(dw, W, w1, tid, nt);
  build_taps(dh, H, h1, tid, nt);
  build_taps(uw, w1, W, tid, nt);
  build_taps(uh, h1, H, tid, nt);
  load_plane<T>(in + plane * H * W, X, H * W, tid, nt);
  __syncthreads();

  // first interpolate call (lp:53): W pass then H pass
  pass_rows(X, W, T1, w1, H, w1, dw, tid, nt);
  __syncthreads();
  pass_cols(T1, w1, w1, h1, dh, tid, nt, [&](int o, int c, float v) { T2[o * w1 + c] = round_mid ? rbf(v) : v; });
  __syncthreads();
  // second interpolate call (lp:54): W pass (into the dead X region) then H pass straight to HBM
  float* T3 = X;
  pass_rows(T2, w1, T3, W, h1, W, uw, tid, nt);
  __syncthreads();
  T* optr = out + plane * H * W;
  pass_cols(T3, W, W, H, uh, tid, nt, [&](int y, int x, float v) { store_from_float<T>(optr, (int64_t)y * W + x, v); });
}

static size_t down_up_lds_bytes(int H, int W, int h1, int w1) {
  size_t fl = (size_t)H * W + (size_t)H * w1 + (size_t)h1 * w1;
  size_t b = fl * 4;
  auto tab = [](int n_out, int taps) { return (size_t)n_out * 8 + (size_t)n_out * taps * 4; };
  b += tab(w1, aa_max_taps(W, w1)) + tab(h1, aa_max_taps(H, h1)) + tab(W, aa_max_taps(w1, W)) +
       tab(H, aa_max_taps(h1, H));
  return (b + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------------------------
// gaussian blur
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

template <typename T>
__global__ __launch_bounds__(1024) void gaussian_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                       int ksize, float sigma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t plane = blockIdx.x;
  float* X = (float*)smem;
  float* Tm = X + (size_t)H * W;
  float* g = Tm + (size_t)H * W;  // [ksize]
  const int pad = ksize / 2;
  // g = exp(-0.5 (x/sigma)^2), x = -(k-1)/2 + j (exact for odd k), normalised by the sequential sum
  if (tid < ksize) {
    float x = (float)tid - 0.5f * (float)(ksize - 1);
    float q = __fdiv_rn(x, sigma);
    g[tid] = expf(__fmul_rn(-0.5f, __fmul_rn(q, q)));
  }
  load_plane<T>(in + plane * H * W, X, H * W, tid, nt);
  __syncthreads();
  float tot = 0.0f;
  for (int j = 0; j < ksize; ++j) tot = __fadd_rn(tot, g[j]);
  __syncthreads();
  if (tid < ksize) g[tid] = __fdiv_rn(g[tid], tot);
  __syncthreads();
  // thread = one column x, walking a strided set of rows (no per-output division); reflect indexing instead of a
  // padded copy
  const bool fits = W <= nt;
  const int ystep = fits ? nt / W : 1;
  const int y0 = fits ? tid / W : 0;
  // W pass
  for (int x = fits ? tid - y0 * W : tid; x < W; x += nt) {
    if (y0 >= ystep) break;
    const bool inner = x >= pad && x + pad < W;
    for (int y = y0; y < H; y += ystep) {
      const float* row = X + (size_t)y * W;
      float acc = 0.0f;
      if (inner) {
        for (int j = 0; j < ksize; ++j) acc = fmaf(g[j], row[x - pad + j], acc);
      } else {
        for (int j = 0; j < ksize; ++j) acc = fmaf(g[j], row[reflect(x - pad + j, W)], acc);
      }
      Tm[(size_t)y * W + x] = acc;
    }
  }
  __syncthreads();
  // H pass
  T* o = out + plane * H * W;
  for (int x = fits ? tid - y0 * W : tid; x < W; x += nt) {
    if (y0 >= ystep) break;
    for (int y = y0; y < H; y += ystep) {
      float acc = 0.0f;
      if (y >= pad && y + pad < H) {
        const float* col = Tm + (size_t)(y - pad) * W + x;
        for (int i = 0; i < ksize; ++i) acc = fmaf(g[i], col[(size_t)i * W], acc);
      } else {
        for (int i = 0; i < ksize; ++i) acc = fmaf(g[i], Tm[(size_t)reflect(y - pad + i, H) * W + x], acc);
      }
      store_from_float<T>(o, (int64_t)y * W + x, acc);
    }
  }
}

// threads per plane-workgroup.  Measured (rocprofv3, profiles/): one C2 video (208 planes, a single wave of
// workgroups) is latency bound -- 1024 threads per plane: 14 us, 256: 24 us; eight videos (1664 planes) are
// throughput bound and want many small workgroups per CU -- 256 threads: 44 us, 1024: 75 us.
static int plane_threads(int H, int W, int64_t planes) {
  const int n = H * W;
  if (planes > 512) return 256;
  return n >= 4096 ? 1024 : (n >= 1024 ? 512 : 256);
}

// planes beyond the LDS budget: same arithmetic through a global workspace (lowpass_big.hip)
int down_up_big(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
                void* workspace, int64_t workspace_bytes, hipStream_t s);
int gaussian_big(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype,
                 void* workspace, int64_t workspace_bytes, hipStream_t s);
int64_t down_up_big_bytes(int64_t planes, int H, int W, int h1, int w1);
int64_t gaussian_big_bytes(int64_t planes, int H, int W, int ksize);

// bandwidth-shaped kernels for 16-byte-granular planes (lowpass_v2.hip); return 1 = shape not covered
int down_up_v2(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
               const void* tables, hipStream_t s);
int gaussian_v2(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype, hipStream_t s);
// register-blocked kernels for many-plane batches (lowpass_v3.hip); return 1 = shape not covered
int gaussian_v3(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype, hipStream_t s);
int down_up_v3(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
               const void* tables, hipStream_t s);

// ALG_LOWPASS_PATH (all paths bit-identical; the tests walk them): 1 = the one-plane-per-workgroup kernels of this file,
// 4 = LDS-sized planes through the global-memory passes of lowpass_big.hip (2 / 3: see lowpass_v3.hip)
static bool force_v1() { return opt(OPT_LOWPASS_PATH) == 1; }
static bool force_global() { return opt(OPT_LOWPASS_PATH) == 4; }

template <typename K>
static int set_lds_limit(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return ALG_ELIMIT;
  // hipFuncSetAttribute is a per-DEVICE property: the memo is keyed by the current device too (a process that drives
  // two GPUs launched the > 48 KiB kernels on the second one without the opt-in); an unknown device (-1) is never memoised
  static thread_local const void* last_fn = nullptr;
  static thread_local size_t last_bytes = 0;
  static thread_local int last_dev = -2;
  const int dev = current_device_slot();
  if (bytes > 48 * 1024 && !(last_fn == (const void*)kernel && last_bytes >= bytes && dev >= 0 && last_dev == dev)) {
    last_fn = (const void*)kernel;
    last_bytes = bytes;
    last_dev = dev;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", bytes, hipGetErrorString(e));
      return ALG_ELAUNCH;
    }
  }
  return ALG_OK;
}

}  // namespace alg

using namespace alg;

static bool global_path_down_up(int H, int W, int h1, int w1) {
  return down_up_lds_bytes(H, W, h1, w1) > 160 * 1024 || force_global();
}

static size_t gaussian_lds_bytes(int H, int W, int ksize) { return (((size_t)2 * H * W + ksize) * 4 + 15) & ~(size_t)15; }

static bool global_path_gaussian(int H, int W, int ksize) {
  return gaussian_lds_bytes(H, W, ksize) > 160 * 1024 || ksize > 255 || force_global();
}

extern "C" int64_t alg_down_up_workspace_bytes(int64_t planes, int H, int W, int h1, int w1) {
  if (planes <= 0 || H <= 0 || W <= 0 || h1 <= 0 || w1 <= 0 || !global_path_down_up(H, W, h1, w1)) return 0;
  return down_up_big_bytes(planes, H, W, h1, w1);
}

extern "C" int64_t alg_gaussian_blur_workspace_bytes(int64_t planes, int H, int W, int ksize) {
  if (planes <= 0 || H <= 0 || W <= 0 || ksize <= 0 || !global_path_gaussian(H, W, ksize)) return 0;
  return gaussian_big_bytes(planes, H, W, ksize);
}

extern "C" int alg_down_up(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype,
                           int round_intermediate, const void* tables, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  if (planes == 0 && H > 0 && W > 0) return ALG_OK;  // empty batch: nothing to do (pointers may be null)
  if (!in || !out || planes < 0 || H <= 0 || W <= 0 || h1 <= 0 || w1 <= 0) {
    set_error("alg_down_up: bad argument (planes=%lld H=%d W=%d h1=%d w1=%d)", (long long)planes, H, W, h1, w1);
    return ALG_EINVAL;
  }
  if (dtype != ALG_F32 && dtype != ALG_BF16) {
    set_error("alg_down_up: unsupported dtype code %d", dtype);
    return ALG_EINVAL;
  }
  if (in == out) {
    set_error("alg_down_up: in and out must not alias");
    return ALG_EINVAL;
  }
  if (planes == 0) return ALG_OK;
  const size_t lds = down_up_lds_bytes(H, W, h1, w1);
  hipStream_t s = (hipStream_t)stream;
  if (global_path_down_up(H, W, h1, w1))
    return down_up_big(in, out, planes, H, W, h1, w1, dtype, round_intermediate ? 1 : 0, workspace, workspace_bytes, s);
  int rc;
  if (!force_v1() && tables) {   // without the caller's tap tables: the kernels below, which build them in LDS per plane
    if ((uintptr_t)tables & 15) {
      set_error("alg_down_up: tables must be 16-byte aligned");
      return ALG_EINVAL;
    }
    rc = down_up_v3(in, out, planes, H, W, h1, w1, dtype, (dtype == ALG_BF16 && round_intermediate) ? 1 : 0, tables, s);
    if (rc <= 0) return rc;
    rc = down_up_v2(in, out, planes, H, W, h1, w1, dtype, (dtype == ALG_BF16 && round_intermediate) ? 1 : 0, tables, s);
    if (rc <= 0) return rc;
  }
  if (dtype == ALG_F32) {
    rc = set_lds_limit(down_up_kernel<float>, lds);
    if (rc == ALG_OK)
      hipLaunchKernelGGL(down_up_kernel<float>, dim3((unsigned)planes), dim3(plane_threads(H, W, planes)), lds, s,
                         (const float*)in, (float*)out, H, W, h1, w1, 0);
  } else {
    rc = set_lds_limit(down_up_kernel<bf16_t>, lds);
    if (rc == ALG_OK)
      hipLaunchKernelGGL(down_up_kernel<bf16_t>, dim3((unsigned)planes), dim3(plane_threads(H, W, planes)), lds, s,
                         (const bf16_t*)in, (bf16_t*)out, H, W, h1, w1, round_intermediate ? 1 : 0);
  }
  if (rc == ALG_ELIMIT) {
    set_error("alg_down_up: plane %dx%d (->%dx%d) needs %zu B of LDS (> 160 KiB); the LDS-resident kernel "
              "covers latent-sized planes only", H, W, h1, w1, lds);
    return rc;
  }
  if (rc != ALG_OK) return rc;
  return check_launch("alg_down_up");
}

extern "C" int alg_gaussian_blur(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma,
                                 int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  if (planes == 0 && H > 0 && W > 0) return ALG_OK;
  if (!in || !out || planes < 0 || H <= 0 || W <= 0) {
    set_error("alg_gaussian_blur: bad argument (planes=%lld H=%d W=%d)", (long long)planes, H, W);
    return ALG_EINVAL;
  }
  if (ksize <= 0 || (ksize & 1) == 0) {
    set_error("alg_gaussian_blur: kernel size must be odd and positive, got %d", ksize);
    return ALG_EINVAL;
  }
  if (!(sigma > 0.0f)) {
    set_error("alg_gaussian_blur: sigma must be positive, got %g", (double)sigma);
    return ALG_EINVAL;
  }
  if (ksize / 2 >= H || ksize / 2 >= W) {
    set_error("alg_gaussian_blur: reflect padding needs ksize/2 < min(H, W) (ksize=%d H=%d W=%d)", ksize, H, W);
    return ALG_EINVAL;
  }
  if (dtype != ALG_F32 && dtype != ALG_BF16) {
    set_error("alg_gaussian_blur: unsupported dtype code %d", dtype);
    return ALG_EINVAL;
  }
  if (in == out) {
    set_error("alg_gaussian_blur: in and out must not alias");
    return ALG_EINVAL;
  }
  if (planes == 0) return ALG_OK;
  const size_t lds = gaussian_lds_bytes(H, W, ksize);
  hipStream_t s = (hipStream_t)stream;
  // kernels wider than 255 taps (an integer `lp_blur_kernel_size` on pixel-sized planes; reflect padding needs ksize / 2 <
  // min(H, W), so such planes are at least 128 x 128) always take the global-memory passes, which have no tap-count limit
  if (global_path_gaussian(H, W, ksize))
    return gaussian_big(in, out, planes, H, W, ksize, sigma, dtype, workspace, workspace_bytes, s);
  int rc;
  if (!force_v1()) {
    rc = gaussian_v3(in, out, planes, H, W, ksize, sigma, dtype, s);
    if (rc <= 0) return rc;
    rc = gaussian_v2(in, out, planes, H, W, ksize, sigma, dtype, s);
    if (rc <= 0) return rc;
  }
  if (dtype == ALG_F32) {
    rc = set_lds_limit(gaussian_kernel<float>, lds);
    if (rc == ALG_OK)
      hipLaunchKernelGGL(gaussian_kernel<float>, dim3((unsigned)planes), dim3(plane_threads(H, W, planes)), lds, s,
                         (const float*)in, (float*)out, H, W, ksize, sigma);
  } else {
    rc = set_lds_limit(gaussian_kernel<bf16_t>, lds);
    if (rc == ALG_OK)
      hipLaunchKernelGGL(gaussian_kernel<bf16_t>, dim3((unsigned)planes), dim3(plane_threads(H, W, planes)), lds, s,
                         (const bf16_t*)in, (bf16_t*)out, H, W, ksize, sigma);
  }
  if (rc == ALG_ELIMIT) {
    set_error("alg_gaussian_blur: plane %dx%d needs %zu B of LDS (> 160 KiB)", H, W, lds);
    return rc;
  }
  if (rc != ALG_OK) return rc;
  return check_launch("alg_gaussian_blur");
}
