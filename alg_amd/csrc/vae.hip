// AutoencoderKLCogVideoX decoder (diffusers; call site cog:428-433 `self.vae.decode(latents).sample`): the memory-bound
// kernels around the convolutions.  Activations are channels-last bf16 in two layouts over the same (Hp, Wp) = (H+2, W+2)
// grid, so a 3x3x3 convolution is ONE GEMM launch whose A rows are constant offsets of its C rows (alg_conv_cl_bf16):
//   * padded  [T + 2][Hp][Wp][C]: what a convolution reads.  Borders are zero (Conv3d zero padding), frames 0 and 1 repeat
//     frame 0 (CogVideoXCausalConv3d: the first frame is repeated kernel_size - 1 times);
//   * virtual [T][Hp][Wp][C]: what a convolution writes: row (t, y, x) is the output voxel when y < H and x < W, the other
//     rows are don't-care values (the wrap-around of the constant-offset trick) that no kernel here ever reads.
// The published decoder runs in batches of latent frames (3 first, then 2) and GroupNorm takes its statistics over one
// batch: `first_len` / `seg_len` carry those segments, so the whole video is normalised in one launch with the same numbers.
#include "common.h"

namespace alg {
namespace vae {

__device__ __forceinline__ void unpack8(const uint4 d, float (&f)[8]) {
  f[0] = __uint_as_float(d.x << 16), f[1] = __uint_as_float(d.x & 0xffff0000u);
  f[2] = __uint_as_float(d.y << 16), f[3] = __uint_as_float(d.y & 0xffff0000u);
  f[4] = __uint_as_float(d.z << 16), f[5] = __uint_as_float(d.z & 0xffff0000u);
  f[6] = __uint_as_float(d.w << 16), f[7] = __uint_as_float(d.w & 0xffff0000u);
}

__device__ __forceinline__ int segment_of(const alg_vae_geom& g, int t) {
  return t < g.first_len ? 0 : 1 + (t - g.first_len) / g.seg_len;
}

// latent frame that frame t was upsampled from (CogVideoXUpsample3D compress_time + CogVideoXSpatialNorm3D interpolate)
__device__ __forceinline__ int latent_frame(const alg_vae_geom& g, int t) {
  return g.lat_first_single ? (t == 0 ? 0 : 1 + (t - 1) / g.lat_rate) : t / g.lat_rate;
}

// ---- GroupNorm statistics, deterministic: per-(frame, split) partial sums, then a fixed-order reduction in double ----
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                         const alg_vae_geom g, int nsplit) {
  const int t = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const int chunks = g.C >> 3;     // 16-byte chunks per row
  const int lanes = 256 / chunks;  // rows in flight per block
  const int chunk = tid % chunks, rl = tid / chunks;
  const int HW = g.H * g.W, Wp = g.W + 2;
  const int per = (HW + nsplit - 1) / nsplit;
  const int v0 = sp * per, v1 = min(v0 + per, HW);
  const bf16_t* xf = x + (int64_t)t * (g.H + 2) * Wp * g.C + chunk * 8;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int v = v0 + rl; v < v1; v += lanes) {
    const int y = v / g.W, xx = v - y * g.W;
    float f[8];
    unpack8(*(const uint4*)(xf + (int64_t)(y * Wp + xx) * g.C), f);
    s0 += (f[0] + f[1]) + (f[2] + f[3]);
    q0 += (f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3]);
    s1 += (f[4] + f[5]) + (f[6] + f[7]);
    q1 += (f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]);
  }
  __shared__ float red[256][4];
  red[tid][0] = s0, red[tid][1] = q0, red[tid][2] = s1, red[tid][3] = q1;
  __syncthreads();
  if (tid < 32) {
    const int halves = g.C >> 7;  // 4-channel halves per group: (C / 32) / 4
    float s = 0.f, q = 0.f;
    for (int e = tid * halves; e < (tid + 1) * halves; ++e) {
      const int ch = e >> 1, hf = (e & 1) * 2;
      for (int r = 0; r < lanes; ++r) s += red[r * chunks + ch][hf], q += red[r * chunks + ch][hf + 1];
    }
    float* o = partial + ((int64_t)(t * nsplit + sp) * 32 + tid) * 2;
    o[0] = s, o[1] = q;
  }
}

// one block per segment: thread (grp, lane) sums a fixed strided subset of the partials in double, then a fixed-order
// tree over the 8 lanes of a group -- deterministic, and 8x the memory parallelism of one thread per group
__global__ __launch_bounds__(256) void gn_final_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                       const alg_vae_geom g, int nsplit, float eps) {
  const int seg = blockIdx.x, grp = threadIdx.x >> 3, ln = threadIdx.x & 7;
  const int f0 = seg == 0 ? 0 : g.first_len + (seg - 1) * g.seg_len;
  const int f1 = min(seg == 0 ? g.first_len : f0 + g.seg_len, g.frames);
  const int n_part = (f1 - f0) * nsplit;
  double s = 0.0, q = 0.0;
  for (int i = ln; i < n_part; i += 8) {
    const float* p = partial + ((int64_t)(f0 * nsplit + i) * 32 + grp) * 2;
    s += (double)p[0], q += (double)p[1];
  }
  __shared__ double red[256][2];
  red[threadIdx.x][0] = s, red[threadIdx.x][1] = q;
  __syncthreads();
  if (ln == 0) {
    for (int j = 1; j < 8; ++j) s += red[threadIdx.x + j][0], q += red[threadIdx.x + j][1];
    const double n = (double)(f1 - f0) * g.H * g.W * (g.C >> 5);
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(seg * 32 + grp) * 2] = (float)mean;
    stats[(seg * 32 + grp) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- CogVideoXSpatialNorm3D + SiLU: virtual -> padded ----
// new_f = GroupNorm(f) * conv_y(zq) + conv_b(zq), then the block's nonlinearity; every tensor op of the reference rounds
// to bf16, so this does too.  zyb holds [conv_y(zq) | conv_b(zq)] at LATENT resolution in the latent's padded layout
// (a 1x1x1 convolution commutes with the nearest-neighbour interpolation the reference applies to zq first).
// MODE 2: the spatial norm above; MODE 1: plain GroupNorm (the encoder's norms); MODE 0: no arithmetic, just the
// virtual -> padded re-layout (the encoder's downsamplers convolve the un-normalised activation).
template <int MODE, bool SILU>
__global__ __launch_bounds__(256) void spatial_norm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                           const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta,
                                                           const bf16_t* __restrict__ zyb, bf16_t* __restrict__ out,
                                                           const alg_vae_geom g) {
  const int tp = blockIdx.y, yp = blockIdx.x;  // padded frame / padded row
  const int Hp = g.H + 2, Wp = g.W + 2, chunks = g.C >> 3;
  const int t = max(tp - 2, 0), y = yp - 1;
  bf16_t* orow = out + ((int64_t)tp * Hp + yp) * Wp * g.C;
  const int items = Wp * chunks;
  if (y < 0 || y >= g.H) {
    for (int i = threadIdx.x; i < items; i += 256) *(uint4*)(orow + (int64_t)i * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  const bf16_t* xrow = x + ((int64_t)t * Hp + y) * Wp * g.C;
  const float* st = MODE ? stats + segment_of(g, t) * 64 : nullptr;
  const bf16_t* zrow = nullptr;
  if (MODE == 2) {
    const int lt = latent_frame(g, t), ly = y / g.lat_scale;
    const int lhp = g.lat_h + 2, lwp = g.lat_w + 2;
    zrow = zyb + (((int64_t)(lt + 2) * lhp + ly + 1) * lwp + 1) * (2 * g.C);
  }
  const int gs = g.C >> 5;  // channels per group: 4, 8, 16
  const int csh = __builtin_ctz((unsigned)chunks);  // C is a power of two: no integer division per item
  for (int i = threadIdx.x; i < items; i += 256) {
    const int xp = i >> csh, chunk = i & (chunks - 1);
    const int xx = xp - 1;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (xx >= 0 && xx < g.W) {
      const int c0 = chunk * 8;
      o = *(const uint4*)(xrow + (int64_t)xx * g.C + c0);
      if (MODE) {
        float f[8], ga[8], be[8], zy[8], zb[8];
        unpack8(o, f);
        unpack8(*(const uint4*)(gamma + c0), ga);
        unpack8(*(const uint4*)(beta + c0), be);
        if (MODE == 2) {
          const bf16_t* z = zrow + (int64_t)(xx / g.lat_scale) * (2 * g.C) + c0;
          unpack8(*(const uint4*)z, zy);
          unpack8(*(const uint4*)(z + g.C), zb);
        }
        const int g0 = c0 / gs, g1 = (c0 + 4) / gs;
        const float m0 = st[g0 * 2], r0 = st[g0 * 2 + 1], m1 = st[g1 * 2], r1 = st[g1 * 2 + 1];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = rbf((f[e] - (e < 4 ? m0 : m1)) * (e < 4 ? r0 : r1) * ga[e] + be[e]);
          if (MODE == 2) a = rbf(rbf(a * zy[e]) + zb[e]);
          if (SILU) a = a / (1.0f + __expf(-a));
          v[e] = a;
        }
        o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      }
    }
    *(uint4*)(orow + (int64_t)i * 8) = o;
  }
}

// ---- CogVideoXUpsample3D interpolation: virtual [T][Hp][Wp][C] -> padded [T2][2H+2][2W+2][C] (no time pad: Conv2d) ----
__global__ __launch_bounds__(256) void upsample_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int T2,
                                                       int H, int W, int C, int compress_time, int first_single) {
  const int t2 = blockIdx.y, yp = blockIdx.x;
  const int Hp = H + 2, Wp = W + 2, Hp2 = 2 * H + 2, Wp2 = 2 * W + 2, chunks = C >> 3;
  const int t = !compress_time ? t2 : (first_single ? (t2 == 0 ? 0 : 1 + (t2 - 1) / 2) : t2 / 2);
  bf16_t* orow = out + ((int64_t)t2 * Hp2 + yp) * Wp2 * C;
  const int items = Wp2 * chunks;
  const int y2 = yp - 1;
  const bool yin = y2 >= 0 && y2 < 2 * H;
  const bf16_t* xrow = x + ((int64_t)t * Hp + (yin ? y2 >> 1 : 0)) * Wp * C;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int xp = i / chunks, chunk = i - xp * chunks;
    const int x2 = xp - 1;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (yin && x2 >= 0 && x2 < 2 * W) o = *(const uint4*)(xrow + (int64_t)(x2 >> 1) * C + chunk * 8);
    *(uint4*)(orow + (int64_t)i * 8) = o;
  }
}

// ---- latent ingest: z (c, l, y, x) strided bf16 -> padded [L + 2][h + 2][w + 2][64], channels >= Cz zero, scaled ----
__global__ __launch_bounds__(256) void pack_latent_kernel(const bf16_t* __restrict__ z, int64_t sc, int64_t sl,
                                                          bf16_t* __restrict__ out, int L, int h, int w, int Cz,
                                                          float scale) {
  const int lp = blockIdx.y, yp = blockIdx.x;
  const int hp = h + 2, wp = w + 2;
  const int l = max(lp - 2, 0), y = yp - 1;
  bf16_t* orow = out + ((int64_t)lp * hp + yp) * wp * 64;
  for (int i = threadIdx.x; i < wp * 64; i += 256) {
    const int xp = i >> 6, c = i & 63;
    const int xx = xp - 1;
    bf16_t v = 0;
    if (c < Cz && y >= 0 && y < h && xx >= 0 && xx < w) v = f2bf(bf2f(z[c * sc + l * sl + (int64_t)y * w + xx]) * scale);
    orow[i] = v;
  }
}

// ---- output: virtual [T][Hp][Wp][4] -> NCTHW bf16 [3][T][H][W]  or  THWC uint8 (VideoProcessor.postprocess_video
// "pil" + run:121-125: (x * 0.5 + 0.5).clamp(0, 1) in bf16, (. * 255).round() in fp32) ----
template <bool U8>
__global__ __launch_bounds__(256) void unpack_video_kernel(const bf16_t* __restrict__ x, void* __restrict__ out, int T,
                                                           int H, int W) {
  const int t = blockIdx.y, y = blockIdx.x;
  const int Hp = H + 2, Wp = W + 2;
  const bf16_t* xrow = x + ((int64_t)t * Hp + y) * Wp * 4;
  for (int xx = threadIdx.x; xx < W; xx += 256) {
    const uint2 d = *(const uint2*)(xrow + xx * 4);
    const float f[3] = {__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16)};
    if (U8) {
      uint8_t* o = (uint8_t*)out + (((int64_t)t * H + y) * W + xx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = rbf(rbf(f[c] * 0.5f) + 0.5f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        o[c] = (uint8_t)__float2int_rn(v * 255.0f);
      }
    } else {
      bf16_t* o = (bf16_t*)out;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[(((int64_t)c * T + t) * H + y) * W + xx] = (bf16_t)(c == 0 ? d.x & 0xffffu : c == 1 ? d.x >> 16 : d.y & 0xffffu);
    }
  }
}

// ---- a strided convolution writes rows at its INPUT's pitch: copy the valid region to the standard virtual layout ----
__global__ __launch_bounds__(256) void repitch_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int H, int W,
                                                      int C, int src_rows, int src_wp) {
  const int t = blockIdx.y, y = blockIdx.x, chunks = C >> 3;
  const bf16_t* xrow = x + ((int64_t)t * src_rows + (int64_t)y * src_wp) * C;
  bf16_t* orow = out + ((int64_t)t * (H + 2) + y) * (W + 2) * C;
  for (int i = threadIdx.x; i < W * chunks; i += 256) *(uint4*)(orow + (int64_t)i * 8) = *(const uint4*)(xrow + (int64_t)i * 8);
}

// ---- virtual [T][Hp][Wp][C] -> planes [C][T][H][W] (the encoder's moments) ----
__global__ __launch_bounds__(256) void unpack_planes_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int T,
                                                            int H, int W, int C) {
  const int t = blockIdx.y, y = blockIdx.x;
  const bf16_t* xrow = x + ((int64_t)t * (H + 2) + y) * (W + 2) * C;
  for (int i = threadIdx.x; i < W * C; i += 256) {
    const int c = i / W, xx = i - c * W;
    out[(((int64_t)c * T + t) * H + y) * W + xx] = xrow[(int64_t)xx * C + c];
  }
}

}  // namespace vae
}  // namespace alg

using namespace alg;

static int check_geom(const char* who, const alg_vae_geom* g) {
  if (!g || g->frames <= 0 || g->H <= 0 || g->W <= 0 || g->C < 128 || g->C > 2048 || (g->C & (g->C - 1)) ||
      g->first_len <= 0 || g->seg_len <= 0 || g->lat_rate <= 0 || g->lat_scale <= 0 || g->lat_h <= 0 || g->lat_w <= 0 ||
      (g->H + g->lat_scale - 1) / g->lat_scale > g->lat_h || (g->W + g->lat_scale - 1) / g->lat_scale > g->lat_w) {
    set_error("%s: bad geometry (C must be a power of two in [128, 2048])", who);
    return ALG_EINVAL;
  }
  return ALG_OK;
}

static int gn_splits(const alg_vae_geom* g) {
  int n = (2048 + g->frames - 1) / g->frames;
  const int hw = g->H * g->W;
  if (n > 64) n = 64;
  if (n > (hw + 255) / 256) n = (hw + 255) / 256;
  return n < 1 ? 1 : n;
}

extern "C" int64_t alg_vae_groupnorm_workspace(const alg_vae_geom* g) {
  if (check_geom("alg_vae_groupnorm_workspace", g) != ALG_OK) return -1;
  return (int64_t)g->frames * gn_splits(g) * 32 * 2 * (int64_t)sizeof(float);
}

extern "C" int alg_vae_groupnorm_stats(const void* x, const alg_vae_geom* g, float eps, void* workspace, float* stats,
                                       void* stream) {
  if (int rc = check_geom("alg_vae_groupnorm_stats", g)) return rc;
  if (!x || !workspace || !stats) {
    set_error("alg_vae_groupnorm_stats: null pointer");
    return ALG_EINVAL;
  }
  const int ns = gn_splits(g);
  const int nseg = g->frames <= g->first_len ? 1 : 1 + (g->frames - g->first_len + g->seg_len - 1) / g->seg_len;
  hipLaunchKernelGGL(vae::gn_partial_kernel, dim3(ns, g->frames), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (float*)workspace, *g, ns);
  hipLaunchKernelGGL(vae::gn_final_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, stats,
                     *g, ns, eps);
  return check_launch("alg_vae_groupnorm_stats");
}

extern "C" int alg_vae_spatial_norm(const void* x, const float* stats, const void* gamma, const void* beta,
                                    const void* zyb, void* out, const alg_vae_geom* g, int silu, void* stream) {
  if (int rc = check_geom("alg_vae_spatial_norm", g)) return rc;
  if (!x || !stats || !gamma || !beta || !zyb || !out) {
    set_error("alg_vae_spatial_norm: null pointer");
    return ALG_EINVAL;
  }
  const dim3 grid(g->H + 2, g->frames + 2);
  if (silu)
    hipLaunchKernelGGL((vae::spatial_norm_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       stats, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)zyb, (bf16_t*)out, *g);
  else
    hipLaunchKernelGGL((vae::spatial_norm_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       stats, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)zyb, (bf16_t*)out, *g);
  return check_launch("alg_vae_spatial_norm");
}

extern "C" int alg_vae_group_norm(const void* x, const float* stats, const void* gamma, const void* beta, void* out,
                                  const alg_vae_geom* g, int silu, void* stream) {
  if (int rc = check_geom("alg_vae_group_norm", g)) return rc;
  if (!x || !stats || !gamma || !beta || !out) {
    set_error("alg_vae_group_norm: null pointer");
    return ALG_EINVAL;
  }
  const dim3 grid(g->H + 2, g->frames + 2);
  if (silu)
    hipLaunchKernelGGL((vae::spatial_norm_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       stats, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)nullptr, (bf16_t*)out, *g);
  else
    hipLaunchKernelGGL((vae::spatial_norm_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       stats, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)nullptr, (bf16_t*)out, *g);
  return check_launch("alg_vae_group_norm");
}

extern "C" int alg_vae_pad(const void* x, void* out, int frames, int H, int W, int C, void* stream) {
  if (frames <= 0 || H <= 0 || W <= 0 || C < 8 || (C & (C - 1)) || !x || !out) {
    set_error("alg_vae_pad: bad argument (C must be a power of two >= 8)");
    return ALG_EINVAL;
  }
  alg_vae_geom g = {};
  g.frames = frames, g.H = H, g.W = W, g.C = C;
  hipLaunchKernelGGL((vae::spatial_norm_kernel<0, false>), dim3(H + 2, frames + 2), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const float*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                     (const bf16_t*)nullptr, (bf16_t*)out, g);
  return check_launch("alg_vae_pad");
}

extern "C" int alg_vae_repitch(const void* x, void* out, int frames, int H, int W, int C, int src_rows, int src_wp,
                               void* stream) {
  if (frames <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || src_wp < W || src_rows < H * src_wp || !x || !out) {
    set_error("alg_vae_repitch: bad argument");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(vae::repitch_kernel, dim3(H, frames), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, H, W, C, src_rows, src_wp);
  return check_launch("alg_vae_repitch");
}

extern "C" int alg_vae_unpack_planes(const void* x, void* out, int frames, int H, int W, int C, void* stream) {
  if (frames <= 0 || H <= 0 || W <= 0 || C <= 0 || !x || !out) {
    set_error("alg_vae_unpack_planes: bad argument");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(vae::unpack_planes_kernel, dim3(H, frames), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, frames, H, W, C);
  return check_launch("alg_vae_unpack_planes");
}

extern "C" int alg_vae_upsample(const void* x, void* out, int frames_out, int H, int W, int C, int compress_time,
                                int first_single, void* stream) {
  if (frames_out <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || !x || !out) {
    set_error("alg_vae_upsample: bad argument");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(vae::upsample_kernel, dim3(2 * H + 2, frames_out), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, frames_out, H, W, C, compress_time, first_single);
  return check_launch("alg_vae_upsample");
}

extern "C" int alg_vae_pack_latent(const void* z, int64_t c_stride, int64_t frame_stride, void* out, int frames, int h,
                                   int w, int channels, float scale, void* stream) {
  if (frames <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels > 64 || !z || !out) {
    set_error("alg_vae_pack_latent: bad argument");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(vae::pack_latent_kernel, dim3(h + 2, frames + 2), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)z, c_stride, frame_stride, (bf16_t*)out, frames, h, w, channels, scale);
  return check_launch("alg_vae_pack_latent");
}

extern "C" int alg_vae_unpack_video(const void* x, void* out, int frames, int H, int W, int to_uint8, void* stream) {
  if (frames <= 0 || H <= 0 || W <= 0 || !x || !out) {
    set_error("alg_vae_unpack_video: bad argument");
    return ALG_EINVAL;
  }
  if (to_uint8)
    hipLaunchKernelGGL(vae::unpack_video_kernel<true>, dim3(H, frames), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, out, frames, H, W);
  else
    hipLaunchKernelGGL(vae::unpack_video_kernel<false>, dim3(H, frames), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, out, frames, H, W);
  return check_launch("alg_vae_unpack_video");
}
