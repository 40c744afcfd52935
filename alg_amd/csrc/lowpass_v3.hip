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
// Shapes not covered (W not a multiple of 4, tap counts without an instantiation, halo too large) return 1 and take v2.
#include <algorithm>

#include <math.h>
#include <stdlib.h>

#include "common.h"

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

// acc += (w.y, w.y) * v
__device__ __forceinline__ void pk_fma_bw_hi(v2f& acc, const v2f w, const v2f v) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(v));
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
  const int H = a.H, W = a.W, Q = W >> 2, XS = W + 2 * P4;
  const int n = H * W, nv = n >> 2;
  float* Xp = (float*)smem;
  float* Tm = Xp + (size_t)H * XS;
  float* g = Tm + (size_t)(H + 2 * PAD) * W;

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
#pragma unroll
  for (int k = 0; k < MAXPRE; ++k) {
    const int idx = tid + k * NT, e = idx << 2, y = e / W;
    pre_dst[k] = y * XS + P4 + (e - y * W);
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
      if (tid + k * NT < nv) *(v4f*)(Xp + pre_dst[k]) = Chunk<T>::unpack(pre[k]);
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
        *(v4f*)(dst + (y + PAD) * W) = o;
        if (y >= 1 && y <= PAD) *(v4f*)(dst + (PAD - y) * W) = o;                              // mirrored above row 0
        if (y >= H - 1 - PAD && y <= H - 2) *(v4f*)(dst + (PAD + 2 * (H - 1) - y) * W) = o;    // mirrored below row H - 1
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
          const v4f v = *(const v4f*)(src + min(y0 + s, last_row) * W);   // padded row y0 + s = original row y0 + s - PAD
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
            *(chunk_t*)(op + (size_t)(y0 + r) * W + (q << 2)) =
                Chunk<T>::pack(v4f{acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y});
        yg += dy, q += dq;
        if (q >= Q) q -= Q, ++yg;
      }
    }
    __syncthreads();   // Xp / Tm are free for the next plane
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

static int threads_override() {
  const char* e = getenv("ALG_LOWPASS_V3_THREADS");
  const int v = e ? atoi(e) : 0;
  return (v == 256 || v == 512) ? v : 0;
}

template <typename T, int K, int NT>
static int launch_g(const void* in, void* out, const GArgs& a, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gaussian_v3_kernel<T, K, NT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e));
      return ALG_ELAUNCH;
    }
  }
  const int wgs = std::max(1, std::min((int)((160 * 1024) / lds), 2048 / NT));
  const int64_t slots = (int64_t)num_cus() * wgs;
  const unsigned grid = (unsigned)std::min<int64_t>(a.planes, slots);
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

}  // namespace v3

// Returns ALG_OK when the launch was made, 1 when this shape is not covered (the caller goes on to lowpass_v2.hip).
int gaussian_v3(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype, hipStream_t s) {
  using namespace v3;
  const char* off = getenv("ALG_LOWPASS_V3");
  if (off && off[0] == '0') return 1;
  const size_t esz = dtype == ALG_F32 ? 4 : 2;
  const int pad = ksize / 2, p4 = (pad + 3) & ~3;
  if ((W & 3) || ksize < 3 || ksize > 19 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((size_t)H * W * esz & 15))
    return 1;
  if (pad + 1 >= H || pad + 1 >= W) return 1;                 // the mirrored rows / columns must be distinct from the edge
  if (planes <= 2 * (int64_t)num_cus()) return 1;             // few planes: latency-bound, the plane-per-workgroup kernels win
  const size_t lds = ((size_t)H * (W + 2 * p4) + (size_t)(H + 2 * pad) * W + ksize + 3) / 4 * 16;
  if (lds > 160 * 1024) return 1;
  // thread count: the one that wastes fewer lanes in the last round of the W pass (H * W / 4 items)
  const int items = H * (W >> 2);
  auto waste = [&](int nt) { return (double)((items + nt - 1) / nt * nt) / items; };
  int nt = threads_override();
  if (!nt) nt = waste(512) < waste(256) - 0.02 ? 512 : 256;
  if ((int64_t)nt * MAXPRE * 4 < (int64_t)H * W || (int64_t)nt * MAXHALO < (int64_t)H * 2 * pad) {
    nt = 512;
    if ((int64_t)nt * MAXPRE * 4 < (int64_t)H * W || (int64_t)nt * MAXHALO < (int64_t)H * 2 * pad) return 1;
  }
  GArgs a;
  a.H = H, a.W = W, a.sigma = sigma, a.planes = planes;
  return dtype == ALG_F32 ? dispatch_g<float>(in, out, a, ksize, lds, nt, s)
                          : dispatch_g<bf16_t>(in, out, a, ksize, lds, nt, s);
}

}  // namespace alg
