// ALG low-pass filters, register-blocked variant (second step).  lowpass_v2.hip showed the batched
// filters to be INSTRUCTION-ISSUE bound: one LDS read, one index computation and one fma per multiply-accumulate (gaussian,
// Wan 480p: 14 k VALU + 4.8 k LDS wave-instructions per 60 x 104 plane against 1.8 k of minimal arithmetic).  Here a lane
// produces a strip of outputs from a window it holds in registers:
//
//   * taps along W: a lane owns FOUR consecutive outputs; the 4 + 2 * pad inputs they need arrive as 16-byte LDS reads
//     from a row image that already carries the reflected halo, and the outputs are accumulated in PAIRS with
//     v_pk_fma_f32: input k serves out[x] with g[j] and out[x + 1] with g[j - 1], so the pair (g[j - 1], g[j]) times the
//     broadcast input advances both chains, each in its own ascending tap order;
//   * taps along H: a lane owns four consecutive columns (two natural pairs) of R consecutive output rows; every 16-byte
//     row read serves up to R outputs per column, the tap weight is broadcast from a register half;
//   * the reflected halo rows / columns are materialised (by the lanes that hold the mirrored element), so no index is
//     reflected per tap; results leave as 16-byte (fp32) or 8-byte (bf16) stores straight from the accumulators.
//
// Arithmetic is the reference chain of lowpass.hip, operation for operation (acc = 0, then fma over ascending taps): a
// packed fma is the same fused operation per half.  The only extra operations are fma(0, finite, acc) at the ends of a
// pair's chain, which leave acc unchanged (they can turn an all-zero-products result -0 into +0; nothing else).
// Shapes not covered (odd W, plane sizes that are not a multiple of 4 elements, tap counts without an instantiation, halo
// too large) return 1 and take v2.
#include <algorithm>
#include <atomic>

#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "lowpass_tabs.h"

namespace alg {
namespace v3 {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// acc += w * (v.S, v.S): both halves of the result take half S of v
template <int S>
__device__ __forceinline__ void pk_fma_bv(v2f& acc, const v2f w, const v2f v) {
  if constexpr (S == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(v));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(v));
}

// acc += (w.S, w.S) * v
template <int S>
__device__ __forceinline__ void pk_fma_bw(v2f& acc, const v2f w, const v2f v) {
  if constexpr (S == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(v));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(v));
}
__device__ __forceinline__ void pk_fma_bw_hi(v2f& acc, const v2f w, const v2f v) { pk_fma_bw<1>(acc, w, v); }

// (w.S, w.S) * v
template <int S>
__device__ __forceinline__ v2f pk_mul_bw(const v2f w, const v2f v) {
  v2f r;
  if constexpr (S == 0)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(v));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(v));
  return r;
}

template <typename T>
struct Chunk;   // four consecutive elements of T as they travel from HBM
template <>
struct Chunk<float> {
  typedef uint4 type;
  static __device__ __forceinline__ v4f unpack(const uint4 v) {
    return v4f{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  }
  static __device__ __forceinline__ uint4 pack(const v4f v) {
    return uint4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  }
};
template <>
struct Chunk<bf16_t> {
  typedef uint2 type;
  static __device__ __forceinline__ v4f unpack(const uint2 v) {
    return v4f{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
               __uint_as_float(v.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ uint2 pack(const v4f v) { return uint2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)}; }
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// a quad of results to HBM at a 4-byte-aligned element offset; `valid` = 4 or 2 elements
template <typename T>
__device__ __forceinline__ void store_quad(T* p, const v4f v, const bool whole) {
  if constexpr (sizeof(T) == 4) {
    if (whole) {
      *(u32x4_a4*)p = u32x4_a4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    } else {
      *(u32x2_a4*)p = u32x2_a4{__float_as_uint(v.x), __float_as_uint(v.y)};
    }
  } else {
    if (whole) {
      *(u32x2_a4*)p = u32x2_a4{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
    } else {
      *(uint32_t*)p = pack_bf2(v.x, v.y);
    }
  }
}

constexpr int MAXPRE = 8;    // 4-element chunks of the next plane a thread keeps in flight
constexpr int MAXHALO = 4;   // mirrored halo elements a thread fetches per plane

struct GArgs {
  int H, W;
  float sigma;
  int64_t planes;
};

// ---------------------------------------------------------------------------------------------------------------------
// gaussian blur, K taps (odd), reflect padding.  LDS: Xp [H][W + 2 P4] (P4 = pad rounded up to 4: rows stay 16-byte
// aligned and output quad x0 reads padded floats [x0, x0 + 2 P4 + 4)), Tm [H + 2 pad][W] (row pass output with the
// mirrored rows in place), K tap weights.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int K, int NT>
__global__ __launch_bounds__(NT) void gaussian_v3_kernel(const T* __restrict__ in, T* __restrict__ out, const GArgs a) {
  constexpr int PAD = K / 2, P4 = (PAD + 3) & ~3, D = P4 - PAD, NR = (2 * P4 + 4) / 4;
  constexpr int R = 4;   // output rows per lane in the column pass
  typedef typename Chunk<T>::type chunk_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int H = a.H, W = a.W, WS = (W + 3) & ~3, Q = WS >> 2, XS = WS + 2 * P4;   // W is even; W = 2 (mod 4): half a last quad
  const int n = H * W, nv = n >> 2;
  float* Xp = (float*)smem;
  float* Tm = Xp + (size_t)H * XS;
  float* g = Tm + (size_t)(H + 2 * PAD) * WS;

  // g = exp(-0.5 (x/sigma)^2), x = -(k-1)/2 + j, normalised by the sequential sum (lowpass.hip gaussian_kernel)
  for (int j = tid; j < K; j += NT) {
    float x = (float)j - 0.5f * (float)(K - 1);
    float q = __fdiv_rn(x, a.sigma);
    g[j] = expf(__fmul_rn(-0.5f, __fmul_rn(q, q)));
  }
  __syncthreads();
  float tot = 0.0f;
  for (int j = 0; j < K; ++j) tot = __fadd_rn(tot, g[j]);
  // wp[m] = (g[m - 1], g[m]), g[-1] = g[K] = 0
  v2f wp[K + 1];
#pragma unroll
  for (int m = 0; m <= K; ++m) {
    wp[m].x = m > 0 ? __fdiv_rn(g[m - 1], tot) : 0.0f;
    wp[m].y = m < K ? __fdiv_rn(g[m], tot) : 0.0f;
  }

  // per-thread constants of the plane walk: where its chunks and halo elements sit (the same for every plane)
  int pre_dst[MAXPRE];
  {
    int y = (tid << 2) / W, c = (tid << 2) - y * W;
    const int sy = (4 * NT) / W, sc = 4 * NT - sy * W;   // one division per thread; chunk k + 1 is 4 NT elements further
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k) {
      pre_dst[k] = (y * XS + P4 + c) | (c + 2 >= W ? 1 : 0);   // bit 0: the chunk's second pair starts the next row
      y += sy, c += sc;
      if (c >= W) c -= W, ++y;
    }
  }
  const int n_halo = H * 2 * PAD;
  int halo_src[MAXHALO], halo_dst[MAXHALO];
#pragma unroll
  for (int k = 0; k < MAXHALO; ++k) {
    const int i = tid + k * NT, y = i / (2 * PAD), t = i - y * 2 * PAD;   // t < PAD: left halo, else right
    const int d = t < PAD ? t + 1 : t - PAD + 1;                          // distance from the edge element
    halo_src[k] = y * W + (t < PAD ? d : W - 1 - d);
    halo_dst[k] = y * XS + (t < PAD ? P4 - d : P4 + W - 1 + d);
  }

  chunk_t pre[MAXPRE];
  T hpre[MAXHALO];
  auto fetch = [&](const int64_t plane) {
    const chunk_t* gp = (const chunk_t*)(in + plane * n);
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k)
      if (tid + k * NT < nv) pre[k] = gp[tid + k * NT];
#pragma unroll
    for (int k = 0; k < MAXHALO; ++k)
      if (tid + k * NT < n_halo) hpre[k] = in[plane * n + halo_src[k]];
  };

  int64_t plane = blockIdx.x;
  if (plane < a.planes) fetch(plane);
  for (; plane < a.planes; plane += gridDim.x) {
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k)
      if (tid + k * NT < nv) {
        const v4f v = Chunk<T>::unpack(pre[k]);
        const int d0 = pre_dst[k] & ~1, d1 = d0 + 2 + ((pre_dst[k] & 1) ? XS - W : 0);
        *(v2f*)(Xp + d0) = v2f{v.x, v.y};
        *(v2f*)(Xp + d1) = v2f{v.z, v.w};
      }
#pragma unroll
    for (int k = 0; k < MAXHALO; ++k)
      if (tid + k * NT < n_halo) Xp[halo_dst[k]] = load_as_float<T>(&hpre[k], 0);
    __syncthreads();
    if (plane + gridDim.x < a.planes) fetch(plane + gridDim.x);   // in flight during the two passes

    // ---- taps along W: item = (row y, quad q) ----
    {
      int y = tid / Q, q = tid - y * Q;
      const int dy = NT / Q, dq = NT - dy * Q;
      for (; y < H; ) {
        const float* src = Xp + y * XS + (q << 2);
        v4f v[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = *(const v4f*)(src + 4 * i);
        v2f a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};   // (out[x0 + 1], out[x0]), (out[x0 + 3], out[x0 + 2])
#pragma unroll
        for (int e = D; e <= D + K + 2; ++e) {
          const v2f pr = (e & 2) ? v2f{v[e >> 2].z, v[e >> 2].w} : v2f{v[e >> 2].x, v[e >> 2].y};
          if (e <= D + K) {
            if (e & 1) pk_fma_bv<1>(a01, wp[e - D], pr); else pk_fma_bv<0>(a01, wp[e - D], pr);
          }
          if (e >= D + 2) {
            if (e & 1) pk_fma_bv<1>(a23, wp[e - D - 2], pr); else pk_fma_bv<0>(a23, wp[e - D - 2], pr);
          }
        }
        const v4f o = {a01.y, a01.x, a23.y, a23.x};
        float* dst = Tm + (q << 2);
        *(v4f*)(dst + (y + PAD) * WS) = o;
        if (y >= 1 && y <= PAD) *(v4f*)(dst + (PAD - y) * WS) = o;                              // mirrored above row 0
        if (y >= H - 1 - PAD && y <= H - 2) *(v4f*)(dst + (PAD + 2 * (H - 1) - y) * WS) = o;    // mirrored below row H - 1
        y += dy, q += dq;
        if (q >= Q) q -= Q, ++y;
      }
    }
    __syncthreads();
    // ---- taps along H: item = (row group yg, quad q); rows y0 .. y0 + R - 1 of four columns ----
    {
      const int G = (H + R - 1) / R, last_row = H + 2 * PAD - 1;
      int yg = tid / Q, q = tid - yg * Q;
      const int dy = NT / Q, dq = NT - dy * Q;
      T* op = out + plane * n;
      for (; yg < G; ) {
        const int y0 = yg * R;
        const float* src = Tm + (q << 2);
        v2f acc[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][0] = v2f{0.0f, 0.0f}, acc[r][1] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < R + K - 1; ++s) {
          const v4f v = *(const v4f*)(src + min(y0 + s, last_row) * WS);   // padded row y0 + s = original row y0 + s - PAD
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int i = s - r;
            if (i >= 0 && i < K) {
              pk_fma_bw_hi(acc[r][0], wp[i], v2f{v.x, v.y});
              pk_fma_bw_hi(acc[r][1], wp[i], v2f{v.z, v.w});
            }
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (y0 + r < H)
            store_quad<T>(op + (size_t)(y0 + r) * W + (q << 2), v4f{acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y},
                          (q << 2) + 4 <= W);
        yg += dy, q += dq;
        if (q >= Q) q -= Q, ++yg;
      }
    }
    __syncthreads();   // Xp / Tm are free for the next plane
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// down_up: two antialiased bilinear resamples (H x W -> h1 x w1 -> H x W), each a W pass then an H pass (lowpass.hip
// down_up_kernel; tables from the shared device blob).  Row strides in LDS are padded to a multiple of 4 floats (WS, VS) so
// that every quad is one 16-byte access; pad columns and the slack behind the regions hold finite values (zero-filled
// once, afterwards only ever overwritten by results of zero weights), because the tap loops run UNCONDITIONALLY over the
// table's tap count: taps past a row's own count have weight 0 in the table.
//   pass 1 (W, down): a lane owns output column o for good (weights in registers), walks rows; reads are plain LDS words.
//   pass 2 (H, down): item = (output row, column quad); weights come in pairs from an 8-byte-aligned LDS record.
//   pass 3 (W, up, 3 taps): a lane owns a quad of output columns for good (4 x (first tap, 3 weights) in registers).
//   pass 4 (H, up, 3 taps): item = (output row, column quad); one 16-byte record {first row, w0, w1, w2} per output row;
//                           the quad leaves straight for HBM.
// LDS: X [H][WS] (later T3 [h1][WS]) | 16 | T1 [H][VS] | T2 [h1][VS] | 16 | dh_lo [h1] | dh_w [h1][TP] | uh_rec [H][4]
// ---------------------------------------------------------------------------------------------------------------------
struct DArgs {
  int H, W, h1, w1, round_mid;
  int64_t planes;
  v2::Tabs tabs;
};

template <typename T, int TDW, int NT>
__global__ __launch_bounds__(NT) void down_up_v3_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                        const uint32_t* __restrict__ gblob, const DArgs a) {
  typedef typename Chunk<T>::type chunk_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int H = a.H, W = a.W, h1 = a.h1, w1 = a.w1;
  const int WS = (W + 3) & ~3, VS = (w1 + 3) & ~3, Q = WS >> 2, QV = VS >> 2;
  const int n = H * W, nv = n >> 2;
  const int tdh = a.tabs.dh.taps, TP = (tdh + 1) & ~1;
  float* X = (float*)smem;
  float* T1 = X + (size_t)H * WS + 16;
  float* T2 = T1 + (size_t)H * VS;
  int* dh_lo = (int*)(T2 + (size_t)h1 * VS + 16);
  float* dh_w = (float*)(dh_lo + ((h1 + 3) & ~3));
  float* uh_rec = dh_w + (((size_t)h1 * TP + 3) & ~(size_t)3);
  float* T3 = X;
  const int lds_floats = (int)(uh_rec + (size_t)H * 4 - X);

  for (int i = tid; i < lds_floats; i += NT) X[i] = 0.0f;
  __syncthreads();
  {
    const int* gmin = (const int*)gblob + a.tabs.dh.off;
    const float* gw = (const float*)(gmin + 2 * h1);
    for (int o = tid; o < h1; o += NT) dh_lo[o] = gmin[o];
    for (int i = tid; i < h1 * tdh; i += NT) {
      const int o = i / tdh, j = i - o * tdh;
      dh_w[o * TP + j] = gw[i];
    }
    const int* umin = (const int*)gblob + a.tabs.uh.off;
    const float* uw_ = (const float*)(umin + 2 * H);
    for (int y = tid; y < H; y += NT) {
      uh_rec[4 * y] = __int_as_float(umin[y]);
      for (int j = 0; j < 3; ++j) uh_rec[4 * y + 1 + j] = uw_[y * 3 + j];
    }
  }
  // pass 1 constants: output column o = tid % w1 of row group tid / w1
  const int rstep1 = NT / w1, rg1 = tid / w1, o1 = tid - rg1 * w1;
  int xm1;
  float w1r[TDW];
  {
    const int* gmin = (const int*)gblob + a.tabs.dw.off;
    const float* gw = (const float*)(gmin + 2 * w1);
    xm1 = gmin[o1];
#pragma unroll
    for (int j = 0; j < TDW; ++j) w1r[j] = j < a.tabs.dw.taps ? gw[o1 * a.tabs.dw.taps + j] : 0.0f;
  }
  // pass 3 constants: output quad q3 = tid % Q of row group tid / Q
  const int rstep3 = NT / Q, rg3 = tid / Q, q3 = tid - rg3 * Q;
  int xm3[4];
  float w3[4][3];
  {
    const int* gmin = (const int*)gblob + a.tabs.uw.off;
    const float* gw = (const float*)(gmin + 2 * W);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = 4 * q3 + i;
      const bool live = x < W;
      xm3[i] = live ? gmin[x] : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) w3[i][j] = live ? gw[x * 3 + j] : 0.0f;
    }
  }
  // where the thread's chunks land: element pairs (e, e + 1) and (e + 2, e + 3) never straddle a row (W is even); bit 0 of
  // pre_dst says that the second pair starts the next row
  int pre_dst[MAXPRE];
  {
    int y = (tid << 2) / W, c = (tid << 2) - y * W;
    const int sy = (4 * NT) / W, sc = 4 * NT - sy * W;   // one division per thread; chunk k + 1 is 4 NT elements further
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k) {
      pre_dst[k] = (y * WS + c) | (c + 2 >= W ? 1 : 0);
      y += sy, c += sc;
      if (c >= W) c -= W, ++y;
    }
  }
  chunk_t pre[MAXPRE];
  auto fetch = [&](const int64_t plane) {
    const chunk_t* gp = (const chunk_t*)(in + plane * n);
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k)
      if (tid + k * NT < nv) pre[k] = gp[tid + k * NT];
  };
  __syncthreads();

  int64_t plane = blockIdx.x;
  if (plane < a.planes) fetch(plane);
  for (; plane < a.planes; plane += gridDim.x) {
#pragma unroll
    for (int k = 0; k < MAXPRE; ++k)
      if (tid + k * NT < nv) {
        const v4f v = Chunk<T>::unpack(pre[k]);
        const int d0 = pre_dst[k] & ~1, d1 = d0 + 2 + ((pre_dst[k] & 1) ? WS - W : 0);
        *(v2f*)(X + d0) = v2f{v.x, v.y};
        *(v2f*)(X + d1) = v2f{v.z, v.w};
      }
    __syncthreads();
    if (plane + gridDim.x < a.planes) fetch(plane + gridDim.x);   // in flight during the four passes

    // ---- pass 1: T1[r][o] = sum_j X[r][xmin[o] + j] w[o][j] ----
    if (rg1 < rstep1) {
      const float* s0 = X + xm1;
      int r = rg1;
      for (; r + rstep1 < H; r += 2 * rstep1) {   // two rows in flight
        const float* sa = s0 + r * WS;
        const float* sb = sa + rstep1 * WS;
        float va[TDW], vb[TDW];
#pragma unroll
        for (int j = 0; j < TDW; ++j) va[j] = sa[j], vb[j] = sb[j];
        float acca = va[0] * w1r[0], accb = vb[0] * w1r[0];
#pragma unroll
        for (int j = 1; j < TDW; ++j) acca = fmaf(va[j], w1r[j], acca), accb = fmaf(vb[j], w1r[j], accb);
        T1[r * VS + o1] = acca;
        T1[(r + rstep1) * VS + o1] = accb;
      }
      if (r < H) {
        const float* sa = s0 + r * WS;
        float va[TDW];
#pragma unroll
        for (int j = 0; j < TDW; ++j) va[j] = sa[j];
        float acca = va[0] * w1r[0];
#pragma unroll
        for (int j = 1; j < TDW; ++j) acca = fmaf(va[j], w1r[j], acca);
        T1[r * VS + o1] = acca;
      }
    }
    __syncthreads();
    // ---- pass 2: T2[o][c] = sum_j T1[xmin[o] + j][c] w[o][j], item = (o, column quad) ----
    for (int it = tid; it < h1 * QV; it += NT) {
      const int o = it / QV, qc = it - o * QV;
      const int lo = dh_lo[o];
      const float* src = T1 + (qc << 2);
      const float* wrow = dh_w + o * TP;
      v2f a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
      for (int j = 0; j < TP; j += 2) {
        const v2f wv = *(const v2f*)(wrow + j);
        const v4f u0 = *(const v4f*)(src + min(lo + j, H - 1) * VS);
        const v4f u1 = *(const v4f*)(src + min(lo + j + 1, H - 1) * VS);
        pk_fma_bw<0>(a0, wv, v2f{u0.x, u0.y});
        pk_fma_bw<0>(a1, wv, v2f{u0.z, u0.w});
        pk_fma_bw<1>(a0, wv, v2f{u1.x, u1.y});
        pk_fma_bw<1>(a1, wv, v2f{u1.z, u1.w});
      }
      v4f r4 = {a0.x, a0.y, a1.x, a1.y};
      if (a.round_mid) r4 = v4f{rbf(r4.x), rbf(r4.y), rbf(r4.z), rbf(r4.w)};
      *(v4f*)(T2 + o * VS + (qc << 2)) = r4;
    }
    __syncthreads();
    // ---- pass 3: T3[h][x] = sum_j T2[h][xmin[x] + j] w[x][j] (X is dead: T3 lives there) ----
    if (rg3 < rstep3) {
      for (int h = rg3; h < h1; h += rstep3) {
        const float* row = T2 + h * VS;
        v4f o4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* sp = row + xm3[i];
          float acc = sp[0] * w3[i][0];
          acc = fmaf(sp[1], w3[i][1], acc);
          acc = fmaf(sp[2], w3[i][2], acc);
          o4[i] = acc;
        }
        *(v4f*)(T3 + h * WS + (q3 << 2)) = o4;
      }
    }
    __syncthreads();
    // ---- pass 4: out[y][x] = sum_j T3[xmin[y] + j][x] w[y][j], item = (y, column quad) -> HBM ----
    {
      T* op = out + plane * n;
      int y = tid / Q, q = tid - y * Q;
      const int dy = NT / Q, dq = NT - dy * Q;
      for (; y < H; ) {
        const v4f rec = *(const v4f*)(uh_rec + 4 * y);
        const int lo = __float_as_int(rec.x);
        const float* src = T3 + (q << 2);
        const v4f u0 = *(const v4f*)(src + lo * WS);
        const v4f u1 = *(const v4f*)(src + min(lo + 1, h1 - 1) * WS);
        const v4f u2 = *(const v4f*)(src + min(lo + 2, h1 - 1) * WS);
        const v2f w01 = {rec.x, rec.y}, w12 = {rec.z, rec.w};   // (lo, w0), (w1, w2)
        v2f a0 = pk_mul_bw<1>(w01, v2f{u0.x, u0.y}), a1 = pk_mul_bw<1>(w01, v2f{u0.z, u0.w});
        pk_fma_bw<0>(a0, w12, v2f{u1.x, u1.y});
        pk_fma_bw<0>(a1, w12, v2f{u1.z, u1.w});
        pk_fma_bw<1>(a0, w12, v2f{u2.x, u2.y});
        pk_fma_bw<1>(a1, w12, v2f{u2.z, u2.w});
        store_quad<T>(op + (size_t)y * W + (q << 2), v4f{a0.x, a0.y, a1.x, a1.y}, (q << 2) + 4 <= W);
        y += dy, q += dq;
        if (q >= Q) q -= Q, ++y;
      }
    }
    __syncthreads();   // T3 (= X) is free for the next plane
  }
}

static int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// planes a launch must have for these kernels to be used (fewer: the plane-per-workgroup kernels of lowpass.hip put up to
// 1024 threads on a plane and have the lower latency; measured cross-over well below one plane per CU)
static int64_t min_planes() { return opt(OPT_LOWPASS_PATH) == 3 ? 1 : num_cus() / 2; }   // ALG_LOWPASS_PATH=3: at any plane count
static bool v3_off() { return opt(OPT_LOWPASS_PATH) == 2; }                               // ALG_LOWPASS_PATH=2: lowpass_v2.hip

static int wgs_cap() {   // EXPERIMENTS tuning knob: resident workgroups per CU the persistent grid is sized for
  return 1 << 20;
}

// Workgroups of `kernel` a CU really holds (registers AND LDS): the persistent grid must not be larger than that, or the
// surplus workgroups wait for a slot and run their planes as a tail (measured: C2 x 8 videos 26.8 us with five workgroups
// per CU requested, four resident; 22.6 us with four).
template <typename K>
static int resident_wgs(K kernel, int nt, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kernel, nt, lds) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    n = 1;
  }
  return n;
}

static int threads_override() {
  return 0;
}

// per kernel instantiation: raise the LDS limit when needed and ask the runtime how many workgroups a CU holds; both are
// remembered for the largest LDS size seen (first calls may race: they compute the same values)
struct Prep {
  std::atomic<size_t> lds{0};
  std::atomic<int> wgs{0};
  std::atomic<int> dev{-2};   // the device the two values above were set / queried on (both are per-device)
};

template <typename K>
static int prepare(Prep& p, K kernel, int nt, size_t lds, int* wgs) {
  const int dev = current_device_slot();
  if (p.lds.load() != lds || dev < 0 || p.dev.load() != dev) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e));
        return ALG_ELAUNCH;
      }
    }
    p.wgs.store(resident_wgs(kernel, nt, lds));
    p.dev.store(dev);
    p.lds.store(lds);
  }
  *wgs = std::min(std::max(p.wgs.load(), 1), wgs_cap());
  return ALG_OK;
}

template <typename T, int K, int NT>
static int launch_g(const void* in, void* out, const GArgs& a, size_t lds, hipStream_t s) {
  static Prep prep;
  int wgs = 1;
  const int rc = prepare(prep, gaussian_v3_kernel<T, K, NT>, NT, lds, &wgs);
  if (rc != ALG_OK) return rc;
  const unsigned grid = (unsigned)std::min<int64_t>(a.planes, (int64_t)num_cus() * wgs);
  hipLaunchKernelGGL((gaussian_v3_kernel<T, K, NT>), dim3(grid), dim3(NT), lds, s, (const T*)in, (T*)out, a);
  return check_launch("alg_gaussian_blur");
}

template <typename T, int K>
static int launch_g_nt(const void* in, void* out, const GArgs& a, size_t lds, int nt, hipStream_t s) {
  return nt == 512 ? launch_g<T, K, 512>(in, out, a, lds, s) : launch_g<T, K, 256>(in, out, a, lds, s);
}

template <typename T>
static int dispatch_g(const void* in, void* out, const GArgs& a, int ksize, size_t lds, int nt, hipStream_t s) {
  switch (ksize) {
    case 3: return launch_g_nt<T, 3>(in, out, a, lds, nt, s);
    case 5: return launch_g_nt<T, 5>(in, out, a, lds, nt, s);
    case 7: return launch_g_nt<T, 7>(in, out, a, lds, nt, s);
    case 9: return launch_g_nt<T, 9>(in, out, a, lds, nt, s);
    case 11: return launch_g_nt<T, 11>(in, out, a, lds, nt, s);
    case 13: return launch_g_nt<T, 13>(in, out, a, lds, nt, s);
    case 15: return launch_g_nt<T, 15>(in, out, a, lds, nt, s);
    case 17: return launch_g_nt<T, 17>(in, out, a, lds, nt, s);
    case 19: return launch_g_nt<T, 19>(in, out, a, lds, nt, s);
    default: return 1;
  }
}


template <typename T, int TDW, int NT>
static int launch_d(const void* in, void* out, const uint32_t* blob, const DArgs& a, size_t lds, hipStream_t s) {
  static Prep prep;
  int wgs = 1;
  const int rc = prepare(prep, down_up_v3_kernel<T, TDW, NT>, NT, lds, &wgs);
  if (rc != ALG_OK) return rc;
  const unsigned grid = (unsigned)std::min<int64_t>(a.planes, (int64_t)num_cus() * wgs);
  hipLaunchKernelGGL((down_up_v3_kernel<T, TDW, NT>), dim3(grid), dim3(NT), lds, s, (const T*)in, (T*)out, blob, a);
  return check_launch("alg_down_up");
}

template <typename T, int TDW>
static int launch_d_nt(const void* in, void* out, const uint32_t* blob, const DArgs& a, size_t lds, int nt,
                       hipStream_t s) {
  switch (nt) {
    case 256: return launch_d<T, TDW, 256>(in, out, blob, a, lds, s);
    case 512: return launch_d<T, TDW, 512>(in, out, blob, a, lds, s);
    default: return launch_d<T, TDW, 1024>(in, out, blob, a, lds, s);
  }
}

template <typename T>
static int dispatch_d(const void* in, void* out, const uint32_t* blob, const DArgs& a, size_t lds, int nt,
                      hipStream_t s) {
  const int taps = a.tabs.dw.taps;
  if (taps <= 5) return launch_d_nt<T, 5>(in, out, blob, a, lds, nt, s);
  if (taps <= 7) return launch_d_nt<T, 7>(in, out, blob, a, lds, nt, s);
  if (taps <= 9) return launch_d_nt<T, 9>(in, out, blob, a, lds, nt, s);
  return launch_d_nt<T, 11>(in, out, blob, a, lds, nt, s);
}

}  // namespace v3

// Returns ALG_OK when the launch was made, 1 when this shape is not covered (the caller goes on to lowpass_v2.hip).
int gaussian_v3(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype, hipStream_t s) {
  using namespace v3;
  if (v3_off()) return 1;
  const int pad = ksize / 2, p4 = (pad + 3) & ~3;
  if ((W & 1) || ((H * W) & 3) || ksize < 3 || ksize > 19 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return 1;
  if (pad + 1 >= H || pad + 1 >= W) return 1;                 // the mirrored rows / columns must be distinct from the edge
  if (planes < min_planes()) return 1;
  const int ws = (W + 3) & ~3;
  const size_t lds = ((size_t)H * (ws + 2 * p4) + (size_t)(H + 2 * pad) * ws + ksize + 3) / 4 * 16;
  if (lds > 160 * 1024) return 1;
  // thread count (measured, Wan 480p planes): 512 when the launch has more than two planes per CU (16 instead of 8 waves per
  // CU with two workgroups resident: 37 -> 33 us for 8 videos), 256 for a single video (9.8 vs 10.9 us)
  int nt = threads_override();
  if (nt == 1024) nt = 512;
  if (!nt) nt = planes > 2 * (int64_t)num_cus() ? 512 : 256;
  if ((int64_t)nt * MAXPRE * 4 < (int64_t)H * W || (int64_t)nt * MAXHALO < (int64_t)H * 2 * pad) {
    nt = 512;
    if ((int64_t)nt * MAXPRE * 4 < (int64_t)H * W || (int64_t)nt * MAXHALO < (int64_t)H * 2 * pad) return 1;
  }
  GArgs a;
  a.H = H, a.W = W, a.sigma = sigma, a.planes = planes;
  return dtype == ALG_F32 ? dispatch_g<float>(in, out, a, ksize, lds, nt, s)
                          : dispatch_g<bf16_t>(in, out, a, ksize, lds, nt, s);
}


// down_up: returns ALG_OK when the launch was made, 1 when this shape is not covered (the caller goes on to lowpass_v2.hip)
int down_up_v3(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
               const void* tables, hipStream_t s) {
  using namespace v3;
  if (v3_off()) return 1;
  const int n = H * W;
  if ((W & 1) || (n & 3) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return 1;
  if (planes < min_planes()) return 1;
  DArgs a;
  a.H = H, a.W = W, a.h1 = h1, a.w1 = w1, a.round_mid = round_mid, a.planes = planes;
  a.tabs = v2::layout(H, W, h1, w1);
  if (a.tabs.uw.taps != 3 || a.tabs.uh.taps != 3 || a.tabs.dw.taps > 11 || a.tabs.dh.taps > 64) return 1;
  const int WS = (W + 3) & ~3, VS = (w1 + 3) & ~3, TP = (a.tabs.dh.taps + 1) & ~1;
  const size_t floats = (size_t)H * WS + 16 + (size_t)H * VS + (size_t)h1 * VS + 16 + ((h1 + 3) & ~3) +
                        (((size_t)h1 * TP + 3) & ~(size_t)3) + (size_t)H * 4;
  const size_t lds = floats * 4;
  if (lds > 160 * 1024) return 1;
  // thread count (measured): the smallest workgroup whose prefetch registers hold a plane -- several small workgroups per
  // CU overlap each other's barrier-separated passes -- unless LDS admits only one workgroup per CU: then the largest
  int nt = threads_override();
  const int by_lds = (int)((160 * 1024) / lds);
  if (!nt) {
    for (int t : {256, 512, 1024}) {
      if ((int64_t)t * MAXPRE * 4 < n || w1 > t || (WS >> 2) > t) continue;
      if (!nt || by_lds == 1) nt = t;
    }
  }
  if (!nt || (int64_t)nt * MAXPRE * 4 < n || w1 > nt || (WS >> 2) > nt) return 1;
  const uint32_t* blob = (const uint32_t*)tables;
  if (!blob) return 1;
  return dtype == ALG_F32 ? dispatch_d<float>(in, out, blob, a, lds, nt, s)
                          : dispatch_d<bf16_t>(in, out, blob, a, lds, nt, s);
}

}  // namespace alg
