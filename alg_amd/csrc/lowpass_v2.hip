// ALG low-pass filters, bandwidth-shaped variant.  Same arithmetic, operation for operation, as the
// one-plane-per-workgroup kernels of lowpass.hip (bit-identical results: tests/test_gpu_lowpass.py); what changes is how the
// chip is fed when many planes are in flight (8 videos x 208 planes, Wan's 420-plane conditions):
//
//   * a PERSISTENT grid (CUs x workgroups that fit by LDS): a workgroup walks planes blockIdx, blockIdx + grid, ...;
//   * the antialias tap tables are built ONCE per shape by a tiny kernel (alg_lowpass_tables_build: strict fp32, the very
//     expressions of build_taps) into a small blob the caller owns and passes to every call: a workgroup copies it to LDS
//     once, not four tables per plane with ~13 divisions each;
//   * the NEXT plane is already in flight (16-byte loads into registers) while the current one runs its passes, so the
//     ~2 us HBM latency of a plane is hidden behind LDS work instead of heading every plane;
//   * the last pass runs with a wave-uniform output row: its three tap weights and row offsets are scalar loads from the
//     global blob (SGPRs) instead of five LDS reads per output;
//   * results are staged in LDS as the output dtype and leave in 16-byte stores (the per-element 2-byte stores of the
//     column pass were 8x the store instructions).
//
// Planes whose byte size is not a multiple of 16 (or misaligned bases) take the original kernels.
#include <algorithm>

#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "lowpass_tabs.h"

namespace alg {
namespace v2 {

constexpr int REG_TAPS = 12;

// Device builder of the four tables of one shape (alg_lowpass_tables_build): lowpass.hip's build_taps, operation for
// operation (every step one IEEE fp32 operation through the never-fused __f*_rn intrinsics; correctly rounded division),
// written into a blob the CALLER owns -- the library neither allocates nor caches device memory.  One thread per output.
__global__ __launch_bounds__(256) void build_tables_kernel(uint32_t* __restrict__ blob, const Tabs tabs, int H, int W, int h1,
                                                           int w1) {
  const unsigned which = blockIdx.y;   // dw: W -> w1, dh: H -> h1, uw: w1 -> W, uh: h1 -> H
  const Tab t = which == 0 ? tabs.dw : which == 1 ? tabs.dh : which == 2 ? tabs.uw : tabs.uh;
  const int in_size = which == 0 ? W : which == 1 ? H : which == 2 ? w1 : h1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.y == 3 && i < 4) {   // the blob is padded to a multiple of four words
    const int last = tabs.uh.off + tabs.uh.words() + i;
    if (last < tabs.words) blob[last] = 0u;
  }
  const int out_size = t.n_out;
  if (i >= out_size) return;
  int* xmin = (int*)blob + t.off;
  int* xsize = xmin + out_size;
  float* w = (float*)(xsize + out_size) + (size_t)i * t.taps;
  const float scale = __fdiv_rn((float)in_size, (float)out_size);
  const float support = scale >= 1.0f ? scale : 1.0f;
  const float invscale = scale >= 1.0f ? __fdiv_rn(1.0f, scale) : 1.0f;
  const float center = __fmul_rn(scale, (float)i + 0.5f);
  int lo = (int)__fadd_rn(__fsub_rn(center, support), 0.5f);
  lo = lo > 0 ? lo : 0;
  int hi = (int)__fadd_rn(__fadd_rn(center, support), 0.5f);
  hi = hi < in_size ? hi : in_size;
  int n = hi - lo;
  n = n < 0 ? 0 : (n > t.taps ? t.taps : n);
  float tot = 0.0f;
  for (int j = 0; j < n; ++j) {
    float x = fabsf(__fmul_rn(__fadd_rn(__fsub_rn((float)(j + lo), center), 0.5f), invscale));
    const float wj = x < 1.0f ? __fsub_rn(1.0f, x) : 0.0f;
    w[j] = wj;
    tot = __fadd_rn(tot, wj);
  }
  for (int j = 0; j < n; ++j) w[j] = tot != 0.0f ? __fdiv_rn(w[j], tot) : w[j];
  for (int j = n; j < t.taps; ++j) w[j] = 0.0f;
  xmin[i] = lo;
  xsize[i] = n;
}

struct TapView {   // a table inside a word array (LDS copy or the global blob)
  const int* xmin;
  const int* xsize;
  const float* w;
  int taps;
};

__device__ __forceinline__ TapView view(const uint32_t* base, const Tab& t) {
  TapView v;
  v.xmin = (const int*)base + t.off;
  v.xsize = v.xmin + t.n_out;
  v.w = (const float*)(v.xsize + t.n_out);
  v.taps = t.taps;
  return v;
}

// dst[r][o] = sum_j w[o][j] * src[r][xmin[o] + j]: a thread owns one output column (weights in registers) and walks rows.
// The taps of a row are fetched as ONE batch of unconditional LDS reads (reads past the row's last tap stay inside the LDS
// allocation and are discarded by a select: the fma chain is exactly acc = s0 w0, then fma over taps 1 .. n-1), two rows at
// a time, so a thread has ~2 * taps reads in flight instead of a read -> fma -> read chain.
template <int NTAP>
__device__ __forceinline__ float row_dot(const float (&v)[NTAP], const float (&w)[NTAP], int n) {
  float acc = n > 0 ? v[0] * w[0] : 0.0f;
#pragma unroll
  for (int j = 1; j < NTAP; ++j) acc = j < n ? fmaf(v[j], w[j], acc) : acc;
  return acc;
}

template <int NTAP>   // taps <= NTAP: weights and one batch of taps per row in registers
__device__ __forceinline__ void pass_rows_reg(const float* s0, int src_ld, float* dst, int dst_ld, int rows, int o, int n,
                                              const float* wp, int taps, int r0, int rstep) {
  float w[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j) w[j] = j < taps ? wp[j] : 0.0f;
  int r = r0;
  if constexpr (NTAP <= 4) {   // short rows: two in flight (the 12-tap form already has 12 reads outstanding per row)
    for (; r + rstep < rows; r += 2 * rstep) {
      const float* sa = s0 + (size_t)r * src_ld;
      const float* sb = sa + (size_t)rstep * src_ld;
      float va[NTAP], vb[NTAP];
#pragma unroll
      for (int j = 0; j < NTAP; ++j) va[j] = sa[j], vb[j] = sb[j];
      dst[(size_t)r * dst_ld + o] = row_dot<NTAP>(va, w, n);
      dst[(size_t)(r + rstep) * dst_ld + o] = row_dot<NTAP>(vb, w, n);
    }
  }
  for (; r < rows; r += rstep) {
    const float* sa = s0 + (size_t)r * src_ld;
    float va[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) va[j] = sa[j];
    dst[(size_t)r * dst_ld + o] = row_dot<NTAP>(va, w, n);
  }
}

__device__ __forceinline__ void pass_rows(const float* src, int src_ld, float* dst, int dst_ld, int rows, int n_out,
                                          const TapView t, int tid, int nthreads) {
  const bool fits = n_out <= nthreads;
  const int rstep = fits ? nthreads / n_out : 1;
  const int r0 = fits ? tid / n_out : 0;
  for (int o = fits ? tid - r0 * n_out : tid; o < n_out; o += nthreads) {
    if (r0 >= rstep) break;
    const int n = t.xsize[o];
    const float* wp = t.w + (size_t)o * t.taps;
    const float* s0 = src + t.xmin[o];
    if (t.taps <= 4) {
      pass_rows_reg<4>(s0, src_ld, dst, dst_ld, rows, o, n, wp, t.taps, r0, rstep);
    } else if (t.taps <= REG_TAPS) {
      pass_rows_reg<REG_TAPS>(s0, src_ld, dst, dst_ld, rows, o, n, wp, t.taps, r0, rstep);
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

// out(o, c) = sum_j w[o][j] * src[xmin[o] + j][c] for the small middle pass (tables in LDS)
template <typename Store>
__device__ __forceinline__ void pass_cols(const float* src, int src_ld, int cols, int n_out, const TapView t, int tid,
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
__device__ __forceinline__ void put(T* p, int i, float v);
template <>
__device__ __forceinline__ void put<float>(float* p, int i, float v) { p[i] = v; }
template <>
__device__ __forceinline__ void put<bf16_t>(bf16_t* p, int i, float v) { p[i] = f2bf(v); }

// The wide last pass: the output row y is wave-uniform (a wave owns a 64-column strip of the plane and walks rows).  The
// rows' tap parameters (first input row, tap count, <= 3 weights: an upsampling pass has 3 taps) are loaded ONCE per
// workgroup, row r in lane r & 63 of a register set, and reach the scalar unit through v_readlane -- no memory latency per
// row; four rows are in flight per wave (12 LDS reads, then the fma chains).
struct RowParams {   // rows [64 k, 64 k + 64) of the table live in lane (row & 63) of set k (k < 2: up to 128 output rows)
  int lo[2], n[2];
  float w[2][3];
};

__device__ __forceinline__ RowParams load_row_params(const uint32_t* gblob, const Tab tab, int lane) {
  RowParams rp;
  const int* gxmin = (const int*)gblob + tab.off;
  const int* gxsize = gxmin + tab.n_out;
  const float* gw = (const float*)(gxsize + tab.n_out);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = min(lane + 64 * k, tab.n_out - 1);
    rp.lo[k] = gxmin[row], rp.n[k] = gxsize[row];
#pragma unroll
    for (int j = 0; j < 3; ++j) rp.w[k][j] = j < tab.taps ? gw[(size_t)row * tab.taps + j] : 0.0f;
  }
  return rp;
}

template <typename T>
__device__ __forceinline__ void pass_cols_fast(const float* src, int src_ld, int cols, int n_out, const RowParams& rp,
                                               T* dst, int dst_ld, int tid, int nthreads) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthreads >> 6;
  const int strips = (cols + 63) >> 6;
  const bool wide = nwaves < strips;                    // more strips than waves: a wave walks strips, all rows
  const int groups = wide ? 1 : nwaves / strips;
  const int grp = wide ? 0 : wave / strips;
  if (grp >= groups) return;
  auto rl = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
  auto rlf = [](float v, int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };
  for (int strip = wide ? wave : wave % strips; strip < strips; strip += wide ? nwaves : strips) {
    const int c = strip * 64 + lane;
    const bool live = c < cols;
    const float* sc = src + (live ? c : 0);
    for (int o0 = grp; o0 < n_out; o0 += 4 * groups) {
      float v[4][3], w[4][3];
      int nn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = min(o0 + u * groups, n_out - 1), l = o & 63;
        const bool hi = o >= 64;
        const int lo = hi ? rl(rp.lo[1], l) : rl(rp.lo[0], l);
        nn[u] = hi ? rl(rp.n[1], l) : rl(rp.n[0], l);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          w[u][j] = hi ? rlf(rp.w[1][j], l) : rlf(rp.w[0][j], l);
          v[u][j] = sc[(size_t)(lo + j) * src_ld];      // rows past the last tap are inside the LDS allocation
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + u * groups;
        float acc = nn[u] > 0 ? v[u][0] * w[u][0] : 0.0f;
        acc = nn[u] > 1 ? fmaf(v[u][1], w[u][1], acc) : acc;
        acc = nn[u] > 2 ? fmaf(v[u][2], w[u][2], acc) : acc;
        if (live && o < n_out) put<T>(dst, o * dst_ld + c, acc);
      }
    }
  }
}

// generic form (more than 3 taps or more than 128 output rows): taps read from the LDS copy of the table
template <typename T>
__device__ __forceinline__ void pass_cols_uniform(const float* src, int src_ld, int cols, int n_out, const TapView t,
                                                  T* dst, int dst_ld, int tid, int nthreads) {
  pass_cols(src, src_ld, cols, n_out, t, tid, nthreads, [&](int o, int c, float v) { put<T>(dst, o * dst_ld + c, v); });
}

// 16 bytes of T in a register -> fp32 in LDS
template <typename T>
__device__ __forceinline__ void unpack_to_lds(const uint4 v, float* lds) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)lds = *(const float4*)&v;
  } else {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    float4 a, b;
    a.x = __uint_as_float(u[0] << 16), a.y = __uint_as_float(u[0] & 0xffff0000u);
    a.z = __uint_as_float(u[1] << 16), a.w = __uint_as_float(u[1] & 0xffff0000u);
    b.x = __uint_as_float(u[2] << 16), b.y = __uint_as_float(u[2] & 0xffff0000u);
    b.z = __uint_as_float(u[3] << 16), b.w = __uint_as_float(u[3] & 0xffff0000u);
    *(float4*)lds = a;
    *(float4*)(lds + 4) = b;
  }
}

struct DUArgs {
  int H, W, h1, w1, round_mid, a_floats;   // a_floats: size of the X / T3 / staging region in floats
  int64_t planes;
  Tabs tabs;
};

template <typename T, int PRE, int NT, int WPS>   // NT threads, WPS = waves per SIMD the register budget must allow
__global__ __launch_bounds__(NT, WPS) void down_up_v2_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                             const uint32_t* __restrict__ gblob, const DUArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  constexpr int nt = NT;
  const int H = a.H, W = a.W, h1 = a.h1, w1 = a.w1;
  constexpr int V = 16 / sizeof(T);
  const int n = H * W, nv = n / V;
  // LDS: A = X [H*W] fp32, later T3 [h1*W] fp32 | staged output [H*W] of T;  T1 [H*w1];  T2 [h1*w1];  table blob
  float* X = (float*)smem;
  float* T1 = X + a.a_floats;
  float* T2 = T1 + (size_t)H * w1;
  uint32_t* lt = (uint32_t*)(T2 + (((size_t)h1 * w1 + 3) & ~(size_t)3));
  for (int i = tid; i < a.tabs.words; i += nt) lt[i] = gblob[i];
  const TapView dw = view(lt, a.tabs.dw), dh = view(lt, a.tabs.dh), uw = view(lt, a.tabs.uw), uh = view(lt, a.tabs.uh);
  const bool fast4 = a.tabs.uh.taps <= 3 && H <= 128;
  const RowParams rp = load_row_params(gblob, a.tabs.uh, tid & 63);
  float* T3 = X;
  T* O = (T*)(X + (((size_t)h1 * W + 3) & ~(size_t)3));

  uint4 pre[PRE];
  int64_t plane = blockIdx.x;
  if (plane < a.planes) {
    const uint4* g = (const uint4*)(in + plane * n);
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * nt;
      if (idx < nv) pre[k] = g[idx];
    }
  }
  for (; plane < a.planes; plane += gridDim.x) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * nt;
      if (idx < nv) unpack_to_lds<T>(pre[k], X + (size_t)idx * V);
    }
    __syncthreads();
    const int64_t next = plane + gridDim.x;
    if (next < a.planes) {   // in flight during the four passes below
      const uint4* g = (const uint4*)(in + next * n);
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const int idx = tid + k * nt;
        if (idx < nv) pre[k] = g[idx];
      }
    }
    // first interpolate call (lp:53): W pass then H pass
    pass_rows(X, W, T1, w1, H, w1, dw, tid, nt);
    __syncthreads();
    pass_cols(T1, w1, w1, h1, dh, tid, nt, [&](int o, int c, float v) { T2[o * w1 + c] = a.round_mid ? rbf(v) : v; });
    __syncthreads();
    // second interpolate call (lp:54): W pass into the dead X region, H pass into the staging buffer behind it
    pass_rows(T2, w1, T3, W, h1, W, uw, tid, nt);
    __syncthreads();
    if (fast4)
      pass_cols_fast<T>(T3, W, W, H, rp, O, W, tid, nt);
    else
      pass_cols_uniform<T>(T3, W, W, H, uh, O, W, tid, nt);
    __syncthreads();
    uint4* go = (uint4*)(out + plane * n);
    const uint4* lo = (const uint4*)O;
    for (int i = tid; i < nv; i += nt) go[i] = lo[i];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// gaussian blur: reflect-indexed separable passes, taps in registers, same skeleton
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

struct GArgs {
  int H, W, ksize;
  float sigma;
  int64_t planes;
};

template <typename T, int PRE, int NT, int WPS>
__global__ __launch_bounds__(NT, WPS) void gaussian_v2_kernel(const T* __restrict__ in, T* __restrict__ out, const GArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  constexpr int nt = NT;
  const int H = a.H, W = a.W, ksize = a.ksize, pad = ksize / 2;
  constexpr int V = 16 / sizeof(T);
  const int n = H * W, nv = n / V;
  float* X = (float*)smem;                   // [H*W] fp32, later the staged output
  float* Tm = X + (((size_t)n + 3) & ~(size_t)3);
  float* g = Tm + (((size_t)n + 3) & ~(size_t)3);
  T* O = (T*)X;
  // g = exp(-0.5 (x/sigma)^2), x = -(k-1)/2 + j, normalised by the sequential sum (lowpass.hip gaussian_kernel)
  for (int j = tid; j < ksize; j += nt) {
    float x = (float)j - 0.5f * (float)(ksize - 1);
    float q = __fdiv_rn(x, a.sigma);
    g[j] = expf(__fmul_rn(-0.5f, __fmul_rn(q, q)));
  }
  __syncthreads();
  float tot = 0.0f;
  for (int j = 0; j < ksize; ++j) tot = __fadd_rn(tot, g[j]);
  __syncthreads();
  for (int j = tid; j < ksize; j += nt) g[j] = __fdiv_rn(g[j], tot);
  __syncthreads();
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwaves = nt >> 6;
  // tap j sits in lane j of `gl` (and taps 64.. in `gh`): a tap reaches the scalar unit by v_readlane, no memory access
  const float gl = lane < ksize ? g[lane] : 0.0f, gh = lane + 64 < ksize ? g[lane + 64] : 0.0f;
  const bool big_k = ksize > 128;
  auto tap = [&](int j) -> float {
    if (big_k) return g[j];
    const float v = j < 64 ? gl : gh;
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), j & 63));
  };
  const int strips = (W + 63) >> 6;
  // waves >= strips: wave -> (strip, row group); fewer waves than strips: a wave walks strips, all rows
  const bool wide = nwaves < strips;
  const int strip0 = wide ? wave : wave % strips, sstep = wide ? nwaves : strips;
  const int ystep = wide ? 1 : nwaves / strips;
  const int y0 = wide ? 0 : (wave / strips < ystep ? wave / strips : H);   // leftover waves of a partial group idle

  uint4 pre[PRE];
  int64_t plane = blockIdx.x;
  if (plane < a.planes) {
    const uint4* gp = (const uint4*)(in + plane * n);
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * nt;
      if (idx < nv) pre[k] = gp[idx];
    }
  }
  for (; plane < a.planes; plane += gridDim.x) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * nt;
      if (idx < nv) unpack_to_lds<T>(pre[k], X + (size_t)idx * V);
    }
    __syncthreads();
    const int64_t next = plane + gridDim.x;
    if (next < a.planes) {
      const uint4* gp = (const uint4*)(in + next * n);
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const int idx = tid + k * nt;
        if (idx < nv) pre[k] = gp[idx];
      }
    }
    // Both passes: a wave owns a 64-column strip and walks rows FOUR at a time; per tap one (reflected) column / row index
    // serves the four rows, so four independent fma chains are in flight (acc = 0, then fma over ascending taps: the
    // reference order).
    for (int strip = strip0; strip < strips; strip += sstep) {
      const int x = strip * 64 + lane;
      const bool live = x < W;
      const bool inner = x >= pad && x + pad < W;
      for (int yb = y0; yb < H; yb += 4 * ystep) {
        const float* r0 = X + (size_t)min(yb, H - 1) * W;
        const float* r1 = X + (size_t)min(yb + ystep, H - 1) * W;
        const float* r2 = X + (size_t)min(yb + 2 * ystep, H - 1) * W;
        const float* r3 = X + (size_t)min(yb + 3 * ystep, H - 1) * W;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int j = 0; j < ksize; ++j) {
          const int xi = live ? (inner ? x - pad + j : reflect(x - pad + j, W)) : 0;
          const float gj = tap(j);
          a0 = fmaf(gj, r0[xi], a0), a1 = fmaf(gj, r1[xi], a1), a2 = fmaf(gj, r2[xi], a2), a3 = fmaf(gj, r3[xi], a3);
        }
        if (live) {
          Tm[(size_t)yb * W + x] = a0;
          if (yb + ystep < H) Tm[(size_t)(yb + ystep) * W + x] = a1;
          if (yb + 2 * ystep < H) Tm[(size_t)(yb + 2 * ystep) * W + x] = a2;
          if (yb + 3 * ystep < H) Tm[(size_t)(yb + 3 * ystep) * W + x] = a3;
        }
      }
    }
    __syncthreads();
    for (int strip = strip0; strip < strips; strip += sstep) {
      const int x = strip * 64 + lane;
      const bool live = x < W;
      const float* col = Tm + (live ? x : 0);
      for (int yb = y0; yb < H; yb += 4 * ystep) {
        const int ya = min(yb, H - 1), yc = min(yb + ystep, H - 1), yd = min(yb + 2 * ystep, H - 1),
                  ye = min(yb + 3 * ystep, H - 1);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int i = 0; i < ksize; ++i) {
          const float gi = tap(i);
          a0 = fmaf(gi, col[(size_t)reflect(ya - pad + i, H) * W], a0);
          a1 = fmaf(gi, col[(size_t)reflect(yc - pad + i, H) * W], a1);
          a2 = fmaf(gi, col[(size_t)reflect(yd - pad + i, H) * W], a2);
          a3 = fmaf(gi, col[(size_t)reflect(ye - pad + i, H) * W], a3);
        }
        if (live) {   // X is dead after the W pass: the staged output goes there
          put<T>(O, yb * W + x, a0);
          if (yb + ystep < H) put<T>(O, (yb + ystep) * W + x, a1);
          if (yb + 2 * ystep < H) put<T>(O, (yb + 2 * ystep) * W + x, a2);
          if (yb + 3 * ystep < H) put<T>(O, (yb + 3 * ystep) * W + x, a3);
        }
      }
    }
    __syncthreads();
    uint4* go = (uint4*)(out + plane * n);
    const uint4* lo = (const uint4*)O;
    for (int i = tid; i < nv; i += nt) go[i] = lo[i];
    __syncthreads();
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

// Geometry.  Few planes (at most two per CU): a plane's latency is what counts -> as many threads per plane as the
// plane-per-workgroup kernels used.  Many planes: the kernel is latency-bound per workgroup (six barrier-separated phases
// per plane), so throughput comes from WAVES PER CU: the thread count is the smallest of 256 / 512 / 1024 that reaches the
// most resident waves given the LDS the shape needs (all variants are compiled for 8 waves per SIMD: <= 64 VGPRs).
struct Geo {
  int threads, pre;
  unsigned grid;
};

static bool geometry(size_t bytes, int n, int64_t planes, size_t lds, int wps_small, int wps_1024, Geo* g) {
  const int cus = num_cus();
  int best_t = 0, best_waves = -1, best_wgs = 1;
  for (int t : {256, 512, 1024}) {
    const size_t p = (bytes + (size_t)t * 16 - 1) / ((size_t)t * 16);
    if (p > 4) continue;
    int wgs = (int)((160 * 1024) / lds);
    wgs = std::min(std::min(wgs, (t == 1024 ? wps_1024 : wps_small) * 256 / t), 8);   // LDS, registers, hardware slots
    if (wgs < 1) continue;
    // at most two planes per CU: a plane's own latency is the whole run time and the one-plane-per-workgroup kernels of
    // lowpass.hip are as fast or faster (measured: C2 one video 12.9 vs 16.6 us) -> not covered here
    if (planes <= 2 * (int64_t)cus) return false;
    const int waves = wgs * t / 64;
    if (waves > best_waves) best_waves = waves, best_t = t, best_wgs = wgs;
  }
  if (!best_t) return false;
  g->threads = best_t;
  g->pre = (int)((bytes + (size_t)best_t * 16 - 1) / ((size_t)best_t * 16));
  const int64_t slots = (int64_t)cus * best_wgs;
  g->grid = (unsigned)(planes < slots ? planes : slots);
  return true;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", bytes, hipGetErrorString(e));
      return ALG_ELAUNCH;
    }
  }
  return ALG_OK;
}

template <typename T, int PRE, int NT>
static int launch_du(const T* in, T* out, const uint32_t* blob, const DUArgs& a, size_t lds, unsigned grid, hipStream_t s) {
  constexpr int WPS = NT == 1024 ? 4 : 6;   // 76-92 VGPRs: six waves per SIMD for the 256 / 512-thread forms
  int rc = set_lds(down_up_v2_kernel<T, PRE, NT, WPS>, lds);
  if (rc != ALG_OK) return rc;
  hipLaunchKernelGGL((down_up_v2_kernel<T, PRE, NT, WPS>), dim3(grid), dim3(NT), lds, s, in, out, blob, a);
  return check_launch("alg_down_up");
}

template <typename T, int PRE, int NT>
static int launch_g(const T* in, T* out, const GArgs& a, size_t lds, unsigned grid, hipStream_t s) {
  constexpr int WPS = 8;                    // <= 47 VGPRs
  int rc = set_lds(gaussian_v2_kernel<T, PRE, NT, WPS>, lds);
  if (rc != ALG_OK) return rc;
  hipLaunchKernelGGL((gaussian_v2_kernel<T, PRE, NT, WPS>), dim3(grid), dim3(NT), lds, s, in, out, a);
  return check_launch("alg_gaussian_blur");
}

// (threads, prefetch registers) -> instantiation
#define ALG_V2_DISPATCH(FN, T, ...)                                                      \
  switch (geo.threads * 8 + geo.pre) {                                                   \
    case 256 * 8 + 1: return FN<T, 1, 256>(__VA_ARGS__);                                 \
    case 256 * 8 + 2: return FN<T, 2, 256>(__VA_ARGS__);                                 \
    case 256 * 8 + 3: return FN<T, 3, 256>(__VA_ARGS__);                                 \
    case 256 * 8 + 4: return FN<T, 4, 256>(__VA_ARGS__);                                 \
    case 512 * 8 + 1: return FN<T, 1, 512>(__VA_ARGS__);                                 \
    case 512 * 8 + 2: return FN<T, 2, 512>(__VA_ARGS__);                                 \
    case 512 * 8 + 3: return FN<T, 3, 512>(__VA_ARGS__);                                 \
    case 512 * 8 + 4: return FN<T, 4, 512>(__VA_ARGS__);                                 \
    case 1024 * 8 + 1: return FN<T, 1, 1024>(__VA_ARGS__);                               \
    case 1024 * 8 + 2: return FN<T, 2, 1024>(__VA_ARGS__);                               \
    case 1024 * 8 + 3: return FN<T, 3, 1024>(__VA_ARGS__);                               \
    default: return FN<T, 4, 1024>(__VA_ARGS__);                                         \
  }

template <typename T>
static int dispatch_du(const void* in, void* out, const uint32_t* blob, const DUArgs& a, size_t lds, const Geo& geo,
                       hipStream_t s) {
  ALG_V2_DISPATCH(launch_du, T, (const T*)in, (T*)out, blob, a, lds, geo.grid, s)
}

template <typename T>
static int dispatch_g(const void* in, void* out, const GArgs& a, size_t lds, const Geo& geo, hipStream_t s) {
  ALG_V2_DISPATCH(launch_g, T, (const T*)in, (T*)out, a, lds, geo.grid, s)
}

}  // namespace v2

// Returns ALG_OK when the launch was made, 1 when this shape is not covered (the caller falls back to lowpass.hip's kernels).
int down_up_v2(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype, int round_mid,
               const void* tables, hipStream_t s) {
  using namespace v2;
  const size_t esz = dtype == ALG_F32 ? 4 : 2;
  const size_t bytes = (size_t)H * W * esz;
  if ((bytes & 15) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return 1;
  DUArgs a;
  a.H = H, a.W = W, a.h1 = h1, a.w1 = w1, a.round_mid = round_mid, a.planes = planes;
  a.tabs = layout(H, W, h1, w1);
  const size_t stage = ((size_t)h1 * W + 3) / 4 * 4 + (bytes + 3) / 4;          // T3 + staged output, in floats
  const size_t a_fl = (std::max((size_t)H * W, stage) + 3) / 4 * 4;
  a.a_floats = (int)a_fl;
  // + 16 floats of slack: the batched tap reads run up to 11 floats past a row's last tap
  const size_t lds = (a_fl + (size_t)H * w1 + (((size_t)h1 * w1 + 3) & ~(size_t)3) + a.tabs.words + 16) * 4;
  if (lds > 160 * 1024 || a.tabs.dw.taps > 64) return 1;
  if (((size_t)H * w1) & 3) return 1;                                              // keeps T2 / the blob 16-byte aligned
  Geo geo;
  if (!geometry(bytes, H * W, planes, lds, 6, 4, &geo)) return 1;
  const uint32_t* blob = (const uint32_t*)tables;
  if (!blob) return 1;
  return dtype == ALG_F32 ? dispatch_du<float>(in, out, blob, a, lds, geo, s)
                          : dispatch_du<bf16_t>(in, out, blob, a, lds, geo, s);
}

int gaussian_v2(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype, hipStream_t s) {
  using namespace v2;
  const size_t esz = dtype == ALG_F32 ? 4 : 2;
  const size_t bytes = (size_t)H * W * esz;
  if ((bytes & 15) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return 1;
  GArgs a;
  a.H = H, a.W = W, a.ksize = ksize, a.sigma = sigma, a.planes = planes;
  const size_t n4 = ((size_t)H * W + 3) & ~(size_t)3;
  const size_t lds = ((2 * n4 + ksize) * 4 + 15) & ~(size_t)15;
  if (lds > 160 * 1024) return 1;
  Geo geo;
  if (!geometry(bytes, H * W, planes, lds, 8, 8, &geo)) return 1;
  return dtype == ALG_F32 ? dispatch_g<float>(in, out, a, lds, geo, s) : dispatch_g<bf16_t>(in, out, a, lds, geo, s);
}

}  // namespace alg

extern "C" int64_t alg_lowpass_tables_bytes(int H, int W, int h1, int w1) {
  if (H <= 0 || W <= 0 || h1 <= 0 || w1 <= 0) return 0;
  return (int64_t)alg::v2::layout(H, W, h1, w1).words * 4;
}

extern "C" int alg_lowpass_tables_build(void* tables, int64_t bytes, int H, int W, int h1, int w1, void* stream) {
  using namespace alg;
  if (!tables || H <= 0 || W <= 0 || h1 <= 0 || w1 <= 0 || ((uintptr_t)tables & 15)) {
    set_error("alg_lowpass_tables_build: bad argument (tables=%p H=%d W=%d h1=%d w1=%d; 16-byte aligned blob)", tables, H, W,
              h1, w1);
    return ALG_EINVAL;
  }
  const v2::Tabs t = v2::layout(H, W, h1, w1);
  if (bytes < (int64_t)t.words * 4) {
    set_error("alg_lowpass_tables_build: blob of %lld bytes, alg_lowpass_tables_bytes says %lld", (long long)bytes,
              (long long)t.words * 4);
    return ALG_EINVAL;
  }
  const int widest = std::max(std::max(H, W), std::max(h1, w1));
  hipLaunchKernelGGL(v2::build_tables_kernel, dim3((unsigned)((widest + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream,
                     (uint32_t*)tables, t, H, W, h1, w1);
  return check_launch("alg_lowpass_tables_build");
}
