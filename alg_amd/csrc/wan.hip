// HBM-bound kernels of the Wan 2.1 DiT forward (diffusers WanTransformer3DModel, call
// site pipeline_wan_image2video_lowpass.py:910-917).  One wave owns one token row (D = ITERS * 512 elements, 16-byte
// accesses per lane, fp32 statistics held in registers, shuffle reductions).  Rounding points follow the published
// module: the FP32LayerNorm / modulation chain is fp32 with ONE rounding to bf16 at the end; RMSNorm rounds once after the
// normalisation and once after the weight; the rotary product is rounded once (the reference evaluates it in fp64).
#include "common.h"

namespace alg {
namespace wan {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

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

// y = bf16( LN_fp32(x) [* w + b] [* (1 + scale[b]) + shift[b]] )
// q8 != nullptr (fp8 mode): the bf16-rounded row is quantised in registers as alg_quantize_fp8_rows would (amax / 448 per
// row, OCP e4m3) and ONLY the bytes + the row scale are written: the separate quantiser pass over y disappears.
template <int ITERS>
__global__ __launch_bounds__(256) void ln_mod_f32_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                         const float* __restrict__ w, const float* __restrict__ bs,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int64_t mod_bs,
                                                         int64_t total_rows, int rows, float eps,
                                                         uint8_t* __restrict__ q8, float* __restrict__ q8_scale) {
  constexpr int D = ITERS * 512;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int bidx = (int)(row / rows);
  const bf16_t* xr = x + row * D;
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
  const float* sc = scale ? scale + (int64_t)bidx * mod_bs : nullptr;
  const float* sh = shift ? shift + (int64_t)bidx * mod_bs : nullptr;
  bf16_t* yr = y ? y + row * D : nullptr;
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c0 = i * 512 + lane * 8;
    float wv[8], bv[8], o[8];
    if (w) {
      const float4 a = *(const float4*)(w + c0), b4 = *(const float4*)(w + c0 + 4);
      wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w; wv[4] = b4.x; wv[5] = b4.y; wv[6] = b4.z; wv[7] = b4.w;
    }
    if (bs) {
      const float4 a = *(const float4*)(bs + c0), b4 = *(const float4*)(bs + c0 + 4);
      bv[0] = a.x; bv[1] = a.y; bv[2] = a.z; bv[3] = a.w; bv[4] = b4.x; bv[5] = b4.y; bv[6] = b4.z; bv[7] = b4.w;
    }
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, h0 = s0, h1 = s0;
    if (sc) { s0 = *(const float4*)(sc + c0); s1 = *(const float4*)(sc + c0 + 4); }
    if (sh) { h0 = *(const float4*)(sh + c0); h1 = *(const float4*)(sh + c0 + 4); }
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float n = (v[i][k] - mean) * rstd;
      if (w) n = n * wv[k];
      if (bs) n = n + bv[k];
      if (sc) n = n * (1.0f + scv[k]);
      if (sh) n = n + shv[k];
      o[k] = n;
    }
    if (q8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[i][k] = rbf(o[k]);  // the value the bf16 tensor would hold
        amax = fmaxf(amax, fabsf(v[i][k]));
      }
    } else {
      *(uint4*)(yr + c0) = pack8(o);
    }
  }
  if (q8) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
    const float qs = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / qs;
    if (lane == 0) q8_scale[row] = qs;
    uint8_t* qr = q8 + row * D;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fminf(fmaxf(v[i][k] * inv, -448.0f), 448.0f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      *(uint2*)(qr + i * 512 + lane * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
  }
}

// The modulated, non-affine form (norm1 / norm3 of every Wan block: y = LN(x) * (1 + scale[b]) + shift[b], fp32 parameters) and the
// affine, unmodulated form (norm2: y = LN(x) * w + b) over many rows (round 5).  ln_mod_f32_kernel fetches the two fp32 parameter rows for EVERY token row -- 8 D bytes through the vector L1 for 4 D
// bytes of HBM traffic.  Here a workgroup belongs to one batch item and keeps (1 + scale) | shift in the LDS (8 D bytes, loaded once),
// a wave runs rpw consecutive rows with the next row's 16-byte loads in flight, and the launcher sizes rpw so that the call is about one
// resident round of workgroups.  Same operations in the same order per element: bit-identical to ln_mod_f32_kernel (bf16 and e4m3
// outputs).
template <int ITERS>
__global__ __launch_bounds__(256) void ln_mod_f32_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int64_t mod_bs, int rows, float eps, uint8_t* __restrict__ q8,
                                                              float* __restrict__ q8_scale, int rpw, int blocks_per_item,
                                                              int plus_one) {
  constexpr int D = ITERS * 512;
  __shared__ __attribute__((aligned(16))) float prm[2 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bidx = blockIdx.x / blocks_per_item, blk = blockIdx.x - bidx * blocks_per_item;
  for (int i = threadIdx.x * 4; i < D; i += 1024) {
    const float4 a = *(const float4*)(scale + (int64_t)bidx * mod_bs + i);
    // plus_one: the modulation form (n * (1 + scale) + shift); otherwise `scale` / `shift` are the affine weight / bias (n * w + b)
    *(float4*)(prm + i) = plus_one ? make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w) : a;
    *(float4*)(prm + D + i) = *(const float4*)(shift + (int64_t)bidx * mod_bs + i);
  }
  __syncthreads();
  const int r0 = (blk * 4 + wave) * rpw;
  if (r0 >= rows) return;
  const int r1 = r0 + rpw < rows ? r0 + rpw : rows;
  const int64_t row_base = (int64_t)bidx * rows;
  uint4 nxt[ITERS];
  {
    const bf16_t* xr = x + (row_base + r0) * D + lane * 8;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) nxt[i] = *(const uint4*)(xr + i * 512);
  }
  for (int r = r0; r < r1; ++r) {
    const int64_t row = row_base + r;
    float v[ITERS][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      unpack8(nxt[i], v[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[i][k];
    }
    if (r + 1 < r1) {
      const bf16_t* xr = x + (row + 1) * D + lane * 8;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) nxt[i] = *(const uint4*)(xr + i * 512);
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
    bf16_t* yr = y ? y + row * D : nullptr;
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c0 = i * 512 + lane * 8;
      const float4 s0 = *(const float4*)(prm + c0), s1 = *(const float4*)(prm + c0 + 4);
      const float4 h0 = *(const float4*)(prm + D + c0), h1 = *(const float4*)(prm + D + c0 + 4);
      const float s1v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float n = (v[i][k] - mean) * rstd;
        n = n * s1v[k];
        n = n + shv[k];
        o[k] = n;
      }
      if (q8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[i][k] = rbf(o[k]);
          amax = fmaxf(amax, fabsf(v[i][k]));
        }
      } else {
        *(uint4*)(yr + c0) = pack8(o);
      }
    }
    if (q8) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
      const float qs = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
      const float inv = 1.0f / qs;
      if (lane == 0) q8_scale[row] = qs;
      uint8_t* qr = q8 + row * D;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = fminf(fmaxf(v[i][k] * inv, -448.0f), 448.0f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *(uint2*)(qr + i * 512 + lane * 8) = make_uint2((unsigned)lo, (unsigned)hi);
      }
    }
  }
}

// any D (the 1280-wide CLIP tokens of WanImageEmbedding): one wave per row, three strided passes over the row
__global__ __launch_bounds__(256) void ln_mod_f32_generic_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                                 const float* __restrict__ w, const float* __restrict__ bs,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int64_t mod_bs,
                                                                 int64_t total_rows, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int bidx = (int)(row / rows);
  const bf16_t* xr = x + row * D;
  float s = 0.0f;
  for (int c = lane; c < D; c += 64) s += bf2f(xr[c]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.0f;
  for (int c = lane; c < D; c += 64) {
    const float d = bf2f(xr[c]) - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  const float* sc = scale ? scale + (int64_t)bidx * mod_bs : nullptr;
  const float* sh = shift ? shift + (int64_t)bidx * mod_bs : nullptr;
  bf16_t* yr = y + row * D;
  for (int c = lane; c < D; c += 64) {
    float n = (bf2f(xr[c]) - mean) * rstd;
    if (w) n = n * w[c];
    if (bs) n = n + bs[c];
    if (sc) n = n * (1.0f + sc[c]);
    if (sh) n = n + sh[c];
    yr[c] = f2bf(n);
  }
}

// in place: x = rope( bf16( bf16(x * rsqrt(mean(x^2) + eps)) * w ) ), rope over interleaved pairs of each 128-wide head
template <int ITERS>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, int64_t x_rs,
                                                           int64_t total_rows, int rows, float eps) {
  constexpr int D = ITERS * 512;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int tok = (int)(row % rows);
  bf16_t* xr = x + row * x_rs;
  float v[ITERS][8];
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    unpack8(*(const uint4*)(xr + i * 512 + lane * 8), v[i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) q = fmaf(v[i][k], v[i][k], q);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c0 = i * 512 + lane * 8;
    float wv[8], o[8];
    unpack8(*(const uint4*)(w + c0), wv);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = rbf(rbf(v[i][k] * rstd) * wv[k]);
    if (cos_tab) {
      const int j0 = (c0 & 127) >> 1;  // pair index inside the head: 4 pairs per lane
      const float4 cs = *(const float4*)(cos_tab + (int64_t)tok * 64 + j0);
      const float4 sn = *(const float4*)(sin_tab + (int64_t)tok * 64 + j0);
      const float cv[4] = {cs.x, cs.y, cs.z, cs.w}, sv[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = o[2 * p], b = o[2 * p + 1];
        o[2 * p] = a * cv[p] - b * sv[p];
        o[2 * p + 1] = a * sv[p] + b * cv[p];
      }
    }
    *(uint4*)(xr + c0) = pack8(o);
  }
}

// out[l][b][j][d] = table[l][j][d] + float(vec[b][(per_j ? j * D : 0) + d])
__global__ __launch_bounds__(256) void modulation_kernel(const float* __restrict__ table, const bf16_t* __restrict__ vec,
                                                         float* __restrict__ out, int L, int B, int J, int D, int per_j) {
  const int64_t total = (int64_t)L * B * J * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int j = (int)((e / D) % J);
    const int b = (int)((e / ((int64_t)D * J)) % B);
    const int l = (int)(e / ((int64_t)D * J * B));
    out[e] = table[((int64_t)l * J + j) * D + d] + bf2f(vec[(int64_t)b * (per_j ? J * D : D) + (per_j ? j * D : 0) + d]);
  }
}

// out[n][(f, gy, gx)][c*ph*pw + py*pw + px] = in[n][c][f][gy*ph + py][gx*pw + px]; columns >= C*ph*pw are zero
__global__ __launch_bounds__(256) void patchify3d_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int N,
                                                         int C, int F, int H, int W, int ph, int pw, int Kpad) {
  const int gh = H / ph, gw = W / pw;
  const int64_t total = (int64_t)N * F * gh * gw * Kpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(e % Kpad);
    int64_t t = e / Kpad;
    const int gx = (int)(t % gw); t /= gw;
    const int gy = (int)(t % gh); t /= gh;
    const int f = (int)(t % F);
    const int n = (int)(t / F);
    bf16_t v = 0;
    if (kk < C * ph * pw) {
      const int px = kk % pw, py = (kk / pw) % ph, c = kk / (pw * ph);
      v = in[((((int64_t)n * C + c) * F + f) * H + gy * ph + py) * W + gx * pw + px];
    }
    out[e] = v;
  }
}

// out[n][c][f][gy*ph + py][gx*pw + px] = in[n][(f, gy, gx)][(py*pw + px)*C + c]
__global__ __launch_bounds__(256) void unpatchify3d_kernel(const bf16_t* __restrict__ in, int64_t ldin,
                                                           bf16_t* __restrict__ out, int N, int C, int F, int H, int W,
                                                           int ph, int pw, int channel_major) {
  const int gh = H / ph, gw = W / pw;
  const int64_t total = (int64_t)N * C * F * H * W;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int xw = (int)(e % W);
    int64_t t = e / W;
    const int yh = (int)(t % H); t /= H;
    const int f = (int)(t % F); t /= F;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const int gx = xw / pw, px = xw % pw, gy = yh / ph, py = yh % ph;
    const int64_t tok = ((int64_t)n * F + f) * gh * gw + (int64_t)gy * gw + gx;
    out[e] = in[tok * ldin + (channel_major ? c * ph * pw + py * pw + px : (py * pw + px) * C + c)];
  }
}

// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) in fp32: out[n] = [cos(t w_i) | sin(t w_i)]
__global__ void timestep_f32_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int dim) {
  const int half = dim / 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * half) return;
  const int i = e % half, r = e / half;
  const float w = expf(-9.210340371976184f * (float)i / (float)half);
  const float a = t[r] * w;
  out[(int64_t)r * dim + i] = cosf(a);
  out[(int64_t)r * dim + half + i] = sinf(a);
}

// y[m][n] = act( sum_k x[m][k] W[n][k] + b[n] ) in fp32, one wave per output; optional bf16 copies of y and of silu(bf16(y))
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         bf16_t* __restrict__ y_bf, bf16_t* __restrict__ y_silu_bf, int M,
                                                         int N, int K, int act) {
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (int64_t)M * N) return;
  const int n = (int)(o % N), m = (int)(o / N);
  const float* xr = x + (int64_t)m * K;
  const float* wr = W + (int64_t)n * K;
  float acc = 0.0f;
  for (int k = lane; k < K; k += 64) acc = fmaf(xr[k], wr[k], acc);
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (b ? b[n] : 0.0f);
    if (act == 1) v = v / (1.0f + __expf(-v));
    if (y) y[o] = v;
    if (y_bf) y_bf[o] = f2bf(v);
    if (y_silu_bf) {
      const float r = rbf(v);
      y_silu_bf[o] = f2bf(r / (1.0f + __expf(-r)));
    }
  }
}

// exact (erf) GELU in place on a bf16 tensor
__global__ __launch_bounds__(256) void gelu_erf_kernel(bf16_t* __restrict__ x, int64_t numel) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = bf2f(x[e]);
    x[e] = f2bf(0.5f * v * (1.0f + erff(v * 0.7071067811865476f)));
  }
}

}  // namespace wan
}  // namespace alg

using namespace alg;

#define DISPATCH_ITERS(ITERS_EXPR, CALL)                                  \
  switch (ITERS_EXPR) {                                                   \
    case 1: { constexpr int IT = 1; CALL; } break;                        \
    case 2: { constexpr int IT = 2; CALL; } break;                        \
    case 3: { constexpr int IT = 3; CALL; } break;                        \
    case 4: { constexpr int IT = 4; CALL; } break;                        \
    case 6: { constexpr int IT = 6; CALL; } break;                        \
    case 8: { constexpr int IT = 8; CALL; } break;                        \
    case 10: { constexpr int IT = 10; CALL; } break;                      \
    case 12: { constexpr int IT = 12; CALL; } break;                      \
    default: ok = false;                                                  \
  }

static int layernorm_mod_entry(const char* who, const void* x, void* y, void* q8, float* q8_scale, const float* weight,
                               const float* bias, const float* scale, const float* shift, int64_t mod_bstride, int batch,
                               int rows, int D, float eps, void* stream) {
  if (batch < 0 || rows < 0 || D <= 0) {
    set_error("%s: bad shape batch=%d rows=%d D=%d", who, batch, rows, D);
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)batch * rows;
  if (total == 0) return ALG_OK;
  if (!x || (!y && !q8) || (q8 && !q8_scale)) {
    set_error("%s: null pointer", who);
    return ALG_EINVAL;
  }
  const dim3 grid((unsigned)((total + 3) / 4)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  bool ok = D % 512 == 0;
  const bool mod_only = !weight && !bias && scale && shift, affine_only = weight && bias && !scale && !shift;
  const float* p0 = mod_only ? scale : weight;
  const float* p1 = mod_only ? shift : bias;
  if (ok && (mod_only || affine_only) && total >= 4096 && D <= 6144 && (!mod_only || (mod_bstride & 3) == 0) &&
      !(((uintptr_t)p0 | (uintptr_t)p1) & 15)) {
    // the modulated non-affine form over many rows: parameters in the LDS, rpw rows per wave, about one resident round of workgroups
    // (three of these 256-thread workgroups fit a CU: 768 on an MI355X)
    const int64_t resident_wgs = (int64_t)device_cus() * 3;
    const int64_t want = (total + 4 * resident_wgs - 1) / (4 * resident_wgs);
    const int rpw = (int)(want < 8 ? 8 : want);
    const int bpi = (rows + 4 * rpw - 1) / (4 * rpw);
    bool built = true;   // (DISPATCH_ITERS clears `ok` for a width that is not instantiated: then the one-row kernel below takes the call)
    {
      bool ok = true;
      DISPATCH_ITERS(D / 512, hipLaunchKernelGGL(wan::ln_mod_f32_rows_kernel<IT>, dim3((unsigned)(bpi * batch)), blk, 0, s,
                                                 (const bf16_t*)x, (bf16_t*)y, p0, p1, mod_only ? mod_bstride : 0, rows, eps,
                                                 (uint8_t*)q8, q8_scale, rpw, bpi, mod_only ? 1 : 0));
      built = ok;
    }
    if (built) return check_launch(who);
  }
  if (ok) {
    DISPATCH_ITERS(D / 512, hipLaunchKernelGGL(wan::ln_mod_f32_kernel<IT>, grid, blk, 0, s, (const bf16_t*)x, (bf16_t*)y,
                                               weight, bias, scale, shift, mod_bstride, total, rows, eps, (uint8_t*)q8,
                                               q8_scale));
  }
  if (!ok) {
    if (q8) {
      set_error("%s: the fp8 output needs D %% 512 == 0 and D <= 6144 (D=%d)", who, D);
      return ALG_EINVAL;
    }
    hipLaunchKernelGGL(wan::ln_mod_f32_generic_kernel, grid, blk, 0, s, (const bf16_t*)x, (bf16_t*)y, weight, bias, scale,
                       shift, mod_bstride, total, rows, D, eps);
  }
  return check_launch(who);
}

extern "C" int alg_layernorm_mod_f32(const void* x, void* y, const float* weight, const float* bias, const float* scale,
                                     const float* shift, int64_t mod_bstride, int batch, int rows, int D, float eps,
                                     void* stream) {
  return layernorm_mod_entry("alg_layernorm_mod_f32", x, y, nullptr, nullptr, weight, bias, scale, shift, mod_bstride, batch,
                             rows, D, eps, stream);
}

extern "C" int alg_layernorm_mod_f32_fp8(const void* x, void* q8, float* q8_scale, const float* weight, const float* bias,
                                         const float* scale, const float* shift, int64_t mod_bstride, int batch, int rows,
                                         int D, float eps, void* stream) {
  return layernorm_mod_entry("alg_layernorm_mod_f32_fp8", x, nullptr, q8, q8_scale, weight, bias, scale, shift, mod_bstride,
                             batch, rows, D, eps, stream);
}

extern "C" int alg_rmsnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab,
                                int64_t x_rstride, int batch, int rows, int D, float eps, void* stream) {
  if (batch < 0 || rows < 0 || D <= 0 || D % 512 || x_rstride % 8 || (cos_tab && (D % 128 || !sin_tab))) {
    set_error("alg_rmsnorm_rope: bad shape batch=%d rows=%d D=%d stride=%lld", batch, rows, D, (long long)x_rstride);
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)batch * rows;
  if (total == 0) return ALG_OK;
  if (!x || !weight) {
    set_error("alg_rmsnorm_rope: null pointer");
    return ALG_EINVAL;
  }
  const dim3 grid((unsigned)((total + 3) / 4)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  bool ok = true;
  DISPATCH_ITERS(D / 512, hipLaunchKernelGGL(wan::rmsnorm_rope_kernel<IT>, grid, blk, 0, s, (bf16_t*)x,
                                             (const bf16_t*)weight, cos_tab, sin_tab, x_rstride, total, rows, eps));
  if (!ok) {
    set_error("alg_rmsnorm_rope: D=%d is not built (D/512 in {1,2,3,4,6,8,10,12})", D);
    return ALG_ELIMIT;
  }
  return check_launch("alg_rmsnorm_rope");
}

static unsigned grid_for(int64_t total) {
  const int64_t want = (total + 255) / 256;
  return (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

extern "C" int alg_wan_modulation(const float* table, const void* vec, float* out, int layers, int batch, int J, int D,
                                  int vec_per_j, void* stream) {
  if (!table || !vec || !out || layers <= 0 || batch <= 0 || J <= 0 || D <= 0) {
    set_error("alg_wan_modulation: bad argument");
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)layers * batch * J * D;
  hipLaunchKernelGGL(wan::modulation_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, table,
                     (const bf16_t*)vec, out, layers, batch, J, D, vec_per_j);
  return check_launch("alg_wan_modulation");
}

extern "C" int alg_patchify3d(const void* in, void* out, int n, int C, int F, int H, int W, int ph, int pw, int Kpad,
                              void* stream) {
  if (n < 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || H % ph || W % pw || Kpad < C * ph * pw) {
    set_error("alg_patchify3d: bad shape");
    return ALG_EINVAL;
  }
  if (n == 0) return ALG_OK;
  if (!in || !out) {
    set_error("alg_patchify3d: null pointer");
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)n * F * (H / ph) * (W / pw) * Kpad;
  hipLaunchKernelGGL(wan::patchify3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     (bf16_t*)out, n, C, F, H, W, ph, pw, Kpad);
  return check_launch("alg_patchify3d");
}

extern "C" int alg_unpatchify3d(const void* in, int64_t ldin, void* out, int n, int C, int F, int H, int W, int ph, int pw,
                                int channel_major, void* stream) {
  if (n < 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || H % ph || W % pw || ldin < C * ph * pw) {
    set_error("alg_unpatchify3d: bad shape");
    return ALG_EINVAL;
  }
  if (n == 0) return ALG_OK;
  if (!in || !out) {
    set_error("alg_unpatchify3d: null pointer");
    return ALG_EINVAL;
  }
  const int64_t total = (int64_t)n * C * F * H * W;
  hipLaunchKernelGGL(wan::unpatchify3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, ldin, (bf16_t*)out, n, C, F, H, W, ph, pw, channel_major);
  return check_launch("alg_unpatchify3d");
}

extern "C" int alg_timestep_embedding_f32(const float* t, float* out, int n, int dim, void* stream) {
  if (!t || !out || n <= 0 || dim <= 0 || dim % 2) {
    set_error("alg_timestep_embedding_f32: bad argument (n=%d dim=%d)", n, dim);
    return ALG_EINVAL;
  }
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(wan::timestep_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, n,
                     dim);
  return check_launch("alg_timestep_embedding_f32");
}

extern "C" int alg_linear_f32(const float* x, const float* W, const float* b, float* y, void* y_bf16, void* y_silu_bf16,
                              int M, int N, int K, int act, void* stream) {
  if (!x || !W || M <= 0 || N <= 0 || K <= 0 || (!y && !y_bf16 && !y_silu_bf16) || act < 0 || act > 1) {
    set_error("alg_linear_f32: bad argument (M=%d N=%d K=%d)", M, N, K);
    return ALG_EINVAL;
  }
  const int64_t outs = (int64_t)M * N;
  hipLaunchKernelGGL(wan::linear_f32_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, W, b,
                     y, (bf16_t*)y_bf16, (bf16_t*)y_silu_bf16, M, N, K, act);
  return check_launch("alg_linear_f32");
}

extern "C" int alg_gelu_erf(void* x, int64_t numel, void* stream) {
  if (numel < 0) {
    set_error("alg_gelu_erf: bad argument");
    return ALG_EINVAL;
  }
  if (numel == 0) return ALG_OK;
  if (!x) {
    set_error("alg_gelu_erf: null pointer");
    return ALG_EINVAL;
  }
  hipLaunchKernelGGL(wan::gelu_erf_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, numel);
  return check_launch("alg_gelu_erf");
}
