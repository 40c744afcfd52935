// bf16 GEMM for the DiT linears on gfx950 MFMA:  C = R + gate * act(A @ B^T + bias).
//
// Shape of the design (MI355X-first, 64-wide waves):
//   * 256x256 output tile per workgroup, BK = 64, 8 waves as 2(M) x 4(N); each wave owns 128x64 =
//     4x2 tiles of v_mfma_f32_32x32x16_bf16 (128 fp32 accumulators per lane, 32 MFMAs per K-tile).
//   * A and B (both K-contiguous: activations [M][K], nn.Linear weights [N][K]) stream HBM -> LDS with
//     16-byte global_load_lds (no VGPR round trip), double-buffered: 2 x (32 KiB A + 32 KiB B) = 128 KiB.
//   * LDS-DMA writes lane-linear, so the bank swizzle lives on the per-lane SOURCE address: a tile row is
//     128 B = eight 16-B slots; slot s of row r is stored at slot s ^ ((r >> 1) & 7), which makes the
//     ds_read_b128 fragment reads (16 rows per lane group, same logical slot) conflict free.
//   * one barrier per K-tile: wait vmcnt(0) -> barrier -> issue tile t+1's DMA -> MFMA on tile t.
//   * workgroup ids are remapped so each XCD (private 4 MiB L2) walks a contiguous run of tiles, grouped
//     8 M-tiles deep so concurrently resident workgroups share A and B panels.
//   * edge tiles: source rows are clamped (min(row, M-1)), stores are guarded -- no padding contract.
#include "common.h"

namespace alg {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;        // double buffered: 128 KiB
constexpr int GROUP_M = 8;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == ALG_ACT_GELU_TANH) {
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * __frcp_rn(1.0f + __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u));
  }
  if (act == ALG_ACT_SILU) {
    return x * __frcp_rn(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  }
  return x;
}

__device__ __forceinline__ int swap_bits23(int n) {
  return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1);
}

// ACT: ALG_ACT_*; RES: residual (+ optional gate) epilogue.  Compile-time so the 128-accumulator epilogue stays
// fully unrolled with static register indexing.
template <int ACT, bool RES>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16_kernel(const alg_gemm_args p, int m_tiles, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- workgroup -> (batch, m_tile, n_tile): XCD-contiguous, grouped along M ----
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles = m_tiles * n_tiles;
  const int b = wg / tiles;
  int t = wg - b * tiles;
  const int grp = t / (GROUP_M * n_tiles);
  const int first_m = grp * GROUP_M;
  const int gsize = min(m_tiles - first_m, GROUP_M);
  t -= grp * GROUP_M * n_tiles;
  const int m0 = (first_m + t % gsize) * BM;
  const int n0 = (t / gsize) * BN;

  const bf16_t* A = (const bf16_t*)p.A + (int64_t)b * p.strideA;
  const bf16_t* B = (const bf16_t*)p.B + (int64_t)b * p.strideB;

  // ---- per-thread DMA sources: 4 rows of A and 4 rows of B, one 16-B slot each ----
  const int srow = tid >> 3;                          // 0..63 (+64 i)
  const int sslot = (tid & 7) ^ ((tid >> 4) & 7);     // logical slot fetched into physical slot tid&7
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = i * 64 + srow;
    a_src[i] = A + (int64_t)min(m0 + r, p.M - 1) * p.lda + sslot * 8;
    b_src[i] = B + (int64_t)min(n0 + r, p.N - 1) * p.ldb + sslot * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK), (lptr_t)(base + (i * 512 + wave * 64) * 16), 16,
                                       0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + kt * BK),
                                       (lptr_t)(base + TILE_BYTES + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };

  // ---- fragment read offsets ----
  const int l31 = lane & 31, h2 = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  const int a_row_off = (wm * 128 + l31) * 128;   // + mt*32*128
  const int b_row_off = (wn * 64 + l31) * 128;    // + nt*32*128

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* As = smem + (kt & 1) * STAGE_BYTES;
    const char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int so = ((2 * ks + h2) ^ sw) * 16;
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf16x8*)(As + a_row_off + mt * 4096 + so);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bfr[nt] = *(const bf16x8*)(Bs + b_row_off + nt * 4096 + so);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- epilogue: bias, activation, gate, residual, store ----
  const bf16_t* bias = (const bf16_t*)p.bias;
  const bf16_t* R = RES ? (const bf16_t*)p.R + (int64_t)b * p.strideR : nullptr;
  const bf16_t* gate = (RES && p.gate) ? (const bf16_t*)p.gate + (int64_t)b * p.strideGate : nullptr;
  const bool bias_row = p.flags & ALG_GEMM_BIAS_PER_ROW;
  const bool perm = p.flags & ALG_GEMM_PERMUTE_COLS;
  bf16_t* Cb = (bf16_t*)p.C + (int64_t)b * p.strideC;
  const int row_base = m0 + wm * 128 + 4 * h2;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + l31;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : p.N - 1;
    const float bcol = (bias && !bias_row) ? bf2f(bias[colc]) : 0.0f;
    const int ccol = perm ? swap_bits23(colc) : colc;
    float g0 = 1.0f, g1 = 1.0f;
    if (RES && gate) {
      g0 = bf2f(gate[colc]);
      g1 = bf2f(gate[p.N + colc]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row_base + mt * 32 + (e & 3) + 8 * (e >> 2);
        float v = acc[mt][nt][e] + bcol;
        if (bias_row && bias) v += bf2f(bias[min(row, p.M - 1)]);
        v = rbf(v);  // nn.Linear returns a bf16 tensor
        if (ACT != ALG_ACT_NONE) v = rbf(act_apply(v, ACT));
        if (RES) {
          v = rbf((row < p.seg_split ? g0 : g1) * v);
          v = rbf(bf2f(R[(int64_t)min(row, p.M - 1) * p.ldr + colc]) + v);
        }
        if (col_ok && row < p.M) Cb[(int64_t)row * p.ldc + ccol] = f2bf(v);
      }
    }
  }
}

}  // namespace alg

using namespace alg;

extern "C" int alg_gemm_bf16(const alg_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) {
    set_error("alg_gemm_bf16: null argument");
    return ALG_EINVAL;
  }
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) {
    set_error("alg_gemm_bf16: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
    return ALG_EINVAL;
  }
  if (a->K % BK != 0) {
    set_error("alg_gemm_bf16: K=%d must be a multiple of %d", a->K, BK);
    return ALG_EINVAL;
  }
  if (a->lda % 8 || a->ldb % 8 || a->strideA % 8 || a->strideB % 8 || ((uintptr_t)a->A & 15) ||
      ((uintptr_t)a->B & 15)) {
    set_error("alg_gemm_bf16: A/B must be 16-byte aligned with lda/ldb/strides multiples of 8 elements");
    return ALG_EINVAL;
  }
  if (a->act < ALG_ACT_NONE || a->act > ALG_ACT_SILU) {
    set_error("alg_gemm_bf16: unknown activation %d", a->act);
    return ALG_EINVAL;
  }
  if ((a->flags & ALG_GEMM_PERMUTE_COLS) && (a->R || a->gate)) {
    set_error("alg_gemm_bf16: PERMUTE_COLS cannot be combined with residual/gate");
    return ALG_EINVAL;
  }
  if (a->R && a->act != ALG_ACT_NONE) {
    set_error("alg_gemm_bf16: an activation cannot be combined with the residual epilogue");
    return ALG_EINVAL;
  }
  if (a->gate && !a->R) {
    set_error("alg_gemm_bf16: gate needs a residual");
    return ALG_EINVAL;
  }
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[4] = {(const void*)gemm_bf16_kernel<ALG_ACT_NONE, true>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_GELU_TANH, false>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_SILU, false>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_NONE, false>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
      if (e != hipSuccess) {
        set_error("alg_gemm_bf16: hipFuncSetAttribute(%d B LDS): %s", GEMM_LDS, hipGetErrorString(e));
        return ALG_ELAUNCH;
      }
    }
    attr_set = true;
  }
  const int m_tiles = (a->M + BM - 1) / BM, n_tiles = (a->N + BN - 1) / BN;
  const int64_t nwg = (int64_t)m_tiles * n_tiles * a->batch;
  if (nwg > 0x7fffffff) {
    set_error("alg_gemm_bf16: grid too large");
    return ALG_ELIMIT;
  }
  const dim3 grid((unsigned)nwg), block(GEMM_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (a->R) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, true>), grid, block, GEMM_LDS, s, *a, m_tiles, n_tiles);
  } else if (a->act == ALG_ACT_GELU_TANH) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_GELU_TANH, false>), grid, block, GEMM_LDS, s, *a, m_tiles, n_tiles);
  } else if (a->act == ALG_ACT_SILU) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_SILU, false>), grid, block, GEMM_LDS, s, *a, m_tiles, n_tiles);
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, false>), grid, block, GEMM_LDS, s, *a, m_tiles, n_tiles);
  }
  return check_launch("alg_gemm_bf16");
}
