#pragma once
// bf16 GEMM for the DiT linears on gfx950 MFMA:  C = R + gate * act(A @ B^T + bias).
//
// Shape of the design (MI355X-first, 64-wide waves):
//   * 256x256 output tile per workgroup, v_mfma_f32_32x32x16_bf16 issued as (B-fragment, A-fragment), i.e. it produces
//     C^T tiles: a lane then holds 4 CONSECUTIVE output columns of one row per register quad, so the epilogue loads
//     bias/gate/residual and stores C with 8-byte accesses.
//   * A and B (both K-contiguous: activations [M][K], nn.Linear weights [N][K]) stream L2 -> LDS with 16-byte
//     global_load_lds (no VGPR round trip) through 128 KiB of LDS.  Schedules (ALG_GEMM_PIPE, measured in
//     profiles/r1_power_and_issue_rates.txt; all are kept for A/B runs and are covered by the parity tests):
//        PIPE 6 (default): 8 waves, 2-stage BK = 64 buffer staged by half-tiles; the two wave groups that share a SIMD
//                run half a phase apart (ping-pong): 8 MFMAs of one wave cover the partner's fragment reads and DMA
//                issue; counted vmcnt(6), the DMA queue never drains;
//        PIPE 0: 8 waves x 128x64, 2 stages of BK = 64, drain + barrier per K-tile;
//        PIPE 7: 4 waves x 128x128 (one wave per SIMD, 1/3 fewer fragment reads per MFMA), every memory instruction in
//                the issue shadow of an MFMA; 2 stages of BK = 64, the DMA queue drains once per K-tile;
//        PIPE 8: 4 waves x 128x128 on a 4-stage ring of BK = 32 half-tiles ([row][64 B], own swizzle): per stage ONE
//                counted wait + ONE barrier, then 32 MFMAs with the next stage's 16 fragment reads and the DMA of the
//                stage three ahead spread evenly behind them -- the queue never drains (vmcnt(8)), a third fewer
//                fragment reads than the 8-wave layouts, 2 barriers per K = 64 instead of 8; staged epilogue.  Measured
//                (profiles/r2_gemm_pipe8_ablation.txt): 870-950 TFLOP/s against 930-1140 for PIPE 6 -- a BK = 32 stage
//                is fetched as 64-byte row segments, i.e. every 128-byte line travels L2 -> L1 twice; with whole-line
//                requests (timing experiment) it ties PIPE 6, without any DMA it runs 1075-1405.  Opt-in, kept for A/B.
//     Measured and removed from the build (they cost 5 minutes of compile time; DESIGN.md section 4b keeps the numbers):
//     4-stage BK = 32 rings (plain / fragment pipeline across the barrier / asm ds_reads + counted lgkmcnt) and two
//     simpler 4-wave schedules.
//     Each schedule is its own translation unit (gemm_p0/p6/p7.hip) so `make -j` compiles them in parallel.
//   * LDS-DMA writes lane-linear, so the bank swizzle lives on the per-lane SOURCE address; the matching XOR is
//     applied on the ds_read_b128 fragment reads, which are conflict free (measured SQ_LDS_BANK_CONFLICT = 0).
//   * workgroup ids are remapped so each XCD (private 4 MiB L2) walks a contiguous run of tiles, grouped
//     8 M-tiles deep so concurrently resident workgroups share A and B panels.
//   * edge tiles: source rows are clamped (min(row, M-1)), stores are guarded -- no padding contract.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// fp8 GEMM on the block-scaled MFMA (1) or on the two-instruction v_mfma_f32_32x32x16_fp8_fp8 form (0, round 1)
#ifndef ALG_FP8_MX
#define ALG_FP8_MX 1
#endif

namespace alg {

// cache-policy bits of the LDS-DMA loads (aux operand: 1 = sc0, 2 = sc1, 4 = nt), per operand, for A/B builds.  Measured on
// the ping-pong schedule at the C2 shapes: nt on A, B or both: no change (+-0.5 %); sc1: -4 % (B) to -6 % (A).
#ifndef ALG_AUX_A
#define ALG_AUX_A 0
#endif
#ifndef ALG_AUX_B
#define ALG_AUX_B 0
#endif
constexpr int BM = 256, BN = 256;
constexpr int GEMM_LDS = 128 * 1024;
constexpr int GROUP_M = 8;

template <int V>
struct IntC { static constexpr int value = V; };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == ALG_ACT_GELU_TANH) {
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3).  The reciprocal is the hardware's
    // v_rcp_f32 (1 ulp): the correctly rounded one costs ten more instructions per output in an epilogue that is VALU-bound,
    // and the result is rounded to bf16 (2^-9) right after.
    // exp(-2u) = exp2(x (k1 + k2 x^2)), k1 = -2 log2(e) sqrt(2/pi), k2 = 0.044715 k1: three operations for the argument
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, k2 = 0.044715f * k1;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * fmaf(k2, x * x, k1)));
  }
  if (act == ALG_ACT_SILU) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  }
  return x;
}

__device__ __forceinline__ void unpack4(const uint2 v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x << 16);
  f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16);
  f[3] = __uint_as_float(v.y & 0xffff0000u);
}

// ACT: ALG_ACT_*; RES: residual (+ optional gate) epilogue; PIPE: staging pipeline (see header).  All compile-time
// so the 128-accumulator epilogue stays fully unrolled with static register indexing.
// FP8: both operands are OCP e4m3 bytes with one float32 scale per row (A: a_scale[M], B: b_scale[N]); a tile row is
// still 128 bytes (BK = 128), so staging, swizzle and fragment reads are unchanged; every 16-byte fragment feeds two
// v_mfma_f32_32x32x16_fp8_fp8 (its low and high 8 bytes -- A and B use the same k order, so any order is a valid
// contraction order) and the epilogue multiplies the fp32 accumulator by a_scale[row] * b_scale[col].
// CONV: A is a zero-padded channels-last volume and K runs over (tap, channel): k-tile kt reads rows shifted by the
// tap's (dt, dy, dx) -- a constant row offset over the padded grid, so an implicit-GEMM convolution costs one scalar
// offset per k-tile and nothing else (see alg_gemm_args.conv_*).
template <int ACT, bool RES, int PIPE, int WNW = 4, bool FP8 = false, bool CONV = false>
__global__ __launch_bounds__(2 * WNW * 64) void gemm_bf16_kernel(const alg_gemm_args p, int m_tiles, int n_tiles,
                                                                 int group_m) {
  static_assert(!FP8 || PIPE == 6, "the fp8 operands are built on the ping-pong schedule");
  static_assert(!CONV || (PIPE == 6 && !FP8), "the convolution addressing is built on the bf16 ping-pong schedule");
  typedef typename std::conditional<FP8, uint8_t, bf16_t>::type elem_t;
  constexpr int EPS = 16 / (int)sizeof(elem_t);  // elements per 16-byte slot
  constexpr int GEMM_THREADS = 2 * WNW * 64;  // 512 (8 waves, 128x64 each) or 256 (4 waves, 128x128 each)
  constexpr int NT = BN / WNW / 32;           // 32-column MFMA tiles per wave: 2 or 4
  static_assert(PIPE == 0 || PIPE == 6 || PIPE == 7 || PIPE == 8,
                "schedules built: 0 (2-stage ring), 6 (ping-pong), 7 (4-wave interleaved), 8 (4-wave BK = 32 ring)");
  static_assert((PIPE == 7 || PIPE == 8) == (WNW == 2), "PIPE 7 / 8 are the 4-wave layouts, the others run 8 waves");
  constexpr bool BK64 = true;
  constexpr bool PP = PIPE == 6;  // ping-pong: a wave owns 2 x 64 rows (one piece per A half-tile) x 2 x 32 columns
  constexpr int BK = FP8 ? 128 : 64;
  constexpr int NSTAGE = 2;
  constexpr int ROW_BYTES = BK * (int)sizeof(elem_t);   // 128
  constexpr int SLOTS = ROW_BYTES / 16;            // 4 or 8 sixteen-byte slots per tile row
  constexpr int TILE_BYTES = BM * ROW_BYTES;       // one operand tile of one stage
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int LD_PER_OP = TILE_BYTES / (GEMM_THREADS * 16);  // glds instructions per thread per operand: 2 or 4
  constexpr int ROWS_PER_LD = GEMM_THREADS / SLOTS;            // tile rows covered by one glds round: 128 or 64
  static_assert(NSTAGE * STAGE_BYTES == GEMM_LDS, "ring must fill the 128 KiB LDS budget");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;

  // debug-only ablation bits ride in the upper half of group_m (ALG_GEMM_ABLATE: 1 = no DMA after the prologue,
  // 2 = no LDS fragment reads in the PIPE 3 main loop); results are garbage, timing shows what the loop is bound by
  const int abl = group_m >> 16;
  group_m &= 0xffff;
  const int tiles = m_tiles * n_tiles;
  // one output tile; the kernel is persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (see the loop at the end)
  auto do_tile = [&](const int wg) {
  const int b = wg / tiles;
  int t = wg - b * tiles;
  const int grp = t / (group_m * n_tiles);
  const int first_m = grp * group_m;
  const int gsize = min(m_tiles - first_m, group_m);
  t -= grp * group_m * n_tiles;
  const int m0 = (first_m + t % gsize) * BM;
  const int n0 = (t / gsize) * BN;

  const elem_t* A = (const elem_t*)p.A + (int64_t)b * p.strideA;
  const elem_t* B = (const elem_t*)p.B + (int64_t)b * p.strideB;

  // ---- per-thread DMA sources: LD_PER_OP rows of A and of B, one 16-B slot each ----
  // tile row r, logical slot s lives at physical slot s ^ swz(r); the thread that fills physical slot (tid % SLOTS)
  // of row r therefore fetches logical slot (tid % SLOTS) ^ swz(r).  swz(r) only depends on tid (see below).
  const int srow = tid / SLOTS;
  const int sw_src = !BK64 ? ((tid >> 4) & 3) : ((tid >> 4) & 7);   // PIPE1: (r >> 2) & 3, PIPE0: (r >> 1) & 7
  const int sslot = (tid & (SLOTS - 1)) ^ sw_src;
  const elem_t* a_src[LD_PER_OP];
  const elem_t* b_src[LD_PER_OP];
#pragma unroll
  for (int i = 0; i < LD_PER_OP; ++i) {
    const int r = i * ROWS_PER_LD + srow;
    const int am = (abl & 8) ? 0 : m0, bn = (abl & 8) ? 0 : n0;  // ablation bit 8: every tile streams tile (0, 0)
    a_src[i] = A + (int64_t)min(am + r, p.M - 1) * p.lda + sslot * EPS;
    b_src[i] = B + (int64_t)min(bn + r, p.N - 1) * p.ldb + sslot * EPS;
  }
  // element offset of k-tile kt within a row of A (uniform: scalar ALU only)
  auto a_koff = [&](int kt) -> int {
    if constexpr (CONV) {
      const int cl = p.conv_cin_log2 - 6;  // log2(k-tiles per tap)
      const int tap = kt >> cl;
      // taps run (dt, dy, dx) with dx < conv_kw: 3, or 4 when two neighbouring voxels share one A row (see alg_conv_cl_bf16)
      const bool k4 = p.conv_kw == 4;
      const int dt = k4 ? tap / 12 : tap / 9, r9 = tap - dt * (k4 ? 12 : 9);
      const int dy = k4 ? r9 >> 2 : r9 / 3, dx = r9 - dy * (k4 ? 4 : 3);
      return ((dt * p.conv_hpwp + dy * p.conv_wp + dx) << p.conv_cin_log2) + (kt & ((1 << cl) - 1)) * BK;
    } else {
      return kt * BK;
    }
  };
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < LD_PER_OP; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK),
                                       (lptr_t)(base + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + kt * BK),
                                       (lptr_t)(base + TILE_BYTES + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, 0);
    }
  };

  // one DMA instruction of a stage (w = 2*i + operand), so the main loop can spread a stage's DMA between MFMAs
  auto stage_piece = [&](int buf, int kt, int w) {
    char* base = smem + buf * STAGE_BYTES;
    const int i = w >> 1;
    if (w & 1)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + kt * BK),
                                       (lptr_t)(base + TILE_BYTES + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK),
                                       (lptr_t)(base + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, 0);
  };

  // ---- fragment read offsets ----
  const int l31 = lane & 31, h2 = lane >> 5;
  const int sw = !BK64 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
  const int a_row_off = (wm * 128 + l31) * ROW_BYTES;   // + mt*32*ROW_BYTES
  const int b_row_off = (wn * (NT * 32) + l31) * ROW_BYTES;    // + nt*32*ROW_BYTES
  // ping-pong mapping: m-tiles 0,1 sit in A half-tile 0 (block rows wm*64 ..), m-tiles 2,3 in half-tile 1 (128 + wm*64 ..);
  // n-tile 0 in B half-tile 0 (block cols wn*32 ..), n-tile 1 in half-tile 1 (128 + wn*32 ..)
  auto pp_a_row = [&](int mt) { return ((mt >> 1) * 128 + wm * 64 + (mt & 1) * 32 + l31) * ROW_BYTES; };
  auto pp_b_row = [&](int nt) { return (nt * 128 + wn * 32 + l31) * ROW_BYTES; };

  // acc += B-fragment x A-fragment over the 16 bytes both lanes hold (bf16: 8 k values; fp8: 16 k values in two MFMAs)
  auto fma_frag = [](const bf16x8 bfrag, const bf16x8 afrag, f32x16 c) -> f32x16 {
    if constexpr (FP8) {
      typedef long l2 __attribute__((ext_vector_type(2)));
      const l2 bq = __builtin_bit_cast(l2, bfrag), aq = __builtin_bit_cast(l2, afrag);
      c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bq[0], aq[0], c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bq[1], aq[1], c, 0, 0, 0);
    } else {
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag, afrag, c, 0, 0, 0);
    }
  };

  f32x16 acc[4][NT];  // acc[mt][nt] holds the TRANSPOSED 32x32 tile: D[n][m]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // Fragments are double-buffered in registers: the 6 ds_read_b128 of k-step ks+1 are issued BEFORE the 8 MFMAs
  // of k-step ks (pinned with sched_barrier: hipcc otherwise sinks the reads behind the MFMAs to save VGPRs and
  // every k-step then eats a full LDS round trip at s_waitcnt lgkmcnt(0)).
  auto load_frags = [&](const char* As, const char* Bs, int ks, bf16x8 (&af)[4], bf16x8 (&bfr)[NT]) {
    const int so = ((2 * ks + h2) ^ sw) * 16;
    // B fragments first: the MFMA order below (mt outer, nt inner) then needs the reads in exactly issue order, so
    // the counted lgkmcnt waits hipcc emits leave the later reads in flight under the earlier MFMAs
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const bf16x8*)(Bs + b_row_off + nt * 32 * ROW_BYTES + so);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf16x8*)(As + a_row_off + mt * 32 * ROW_BYTES + so);
  };
  auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bfr)[NT]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = fma_frag(bfr[nt], af[mt], acc[mt][nt]);
  };
  auto compute = [&](int buf) {
    const char* As = smem + buf * STAGE_BYTES;
    const char* Bs = As + TILE_BYTES;
    constexpr int KS = BK / 16;
    bf16x8 af0[4], bf0[NT], af1[4], bf1[NT];
    load_frags(As, Bs, 0, af0, bf0);
    // interleave: one ds_read of the NEXT k-step behind each of the first six MFMAs of the current one (an MFMA
    // occupies the matrix pipe for 32 cycles but the wave's issue slot for only 4, so the reads go out under it)
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < 4 + NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - (4 + NT), 0);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      load_frags(As, Bs, ks + 1, af1, bf1);
      mma(af0, bf0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KS) {
        load_frags(As, Bs, ks + 2, af0, bf0);
        mma(af1, bf1);
        interleave();
      } else {
        mma(af1, bf1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nk = p.K / BK;
  if constexpr (PIPE == 0) {
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      compute(kt & 1);
    }
  } else if constexpr (PIPE == 6) {
    // 8-wave ping-pong over half-tiles (the guide's 256^2 8-phase structure, restated for 32x32x16 C^T tiles).
    //   * waves 0-3 (wm = 0) and 4-7 (wm = 1) share SIMDs pairwise and run HALF A PHASE apart (one extra barrier up
    //     front for wm = 1, one at the end for wm = 0): while one group sits in its 8-MFMA section the other issues
    //     its fragment reads and DMA -- the matrix pipe of every SIMD always has exactly one wave feeding it;
    //   * a phase = [ds_reads + one half-tile of DMA] lgkmcnt(0) barrier [8 MFMA = one 64x32 quadrant x K 64] barrier;
    //     4 phases per K-tile: (A0,B0) (A0,B1) (A1,B1) (A1,B0); A0/A1 and B0/B1 come from DIFFERENT half-tiles, so a
    //     K-tile is consumed half-tile by half-tile and restaged the same way, 7 half-tiles ahead:
    //         phase p0 stages B0 of K-tile t+1;  p1: A0 of t+2;  p2: B1 of t+2;  p3: A1 of t+2 and waits vmcnt(6)
    //     (everything of K-tile t+1 has landed, three half-tiles of t+2 stay in flight -- the queue never drains);
    //   * RAW: the wait sits before p3's first barrier, K-tile t+1 is first read in the next phase (after the barrier
    //     both groups have passed).  WAR: every fragment read is retired (lgkmcnt(0)) before the barrier that precedes
    //     the other group's -- and a phase later its own -- restage of that half-tile.
    auto stage_half = [&](int kt, int kind) {  // kind: 0 = A rows 0..127, 1 = A rows 128..255, 2/3 = B likewise
      char* base = smem + (kt & 1) * STAGE_BYTES + (kind >> 1) * TILE_BYTES;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = (kind & 1) * 2 + ii;
        const elem_t* src = (kind >> 1) ? b_src[i] + kt * BK : a_src[i] + a_koff(kt);
        if (kind >> 1)
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, ALG_AUX_B);
        else
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * GEMM_THREADS + wave * 64) * 16), 16, 0, ALG_AUX_A);
      }
    };
    bf16x8 af[2][4], b0f[4], b1f[4];
    auto load_a = [&](const char* As, int half) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          af[m2][ks] = *(const bf16x8*)(As + pp_a_row(half * 2 + m2) + (((2 * ks + h2) ^ sw) * 16));
    };
    auto load_b = [&](const char* Bs, int half, bf16x8 (&bf)[4]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *(const bf16x8*)(Bs + pp_b_row(half) + (((2 * ks + h2) ^ sw) * 16));
    };
    auto quad = [&](int mh, int nt, const bf16x8 (&bf)[4]) {
      __builtin_amdgcn_s_setprio(1);
      if constexpr (FP8 && ALG_FP8_MX) {
        // CDNA4's fp8 rate lives on the block-scaled instruction only (v_mfma_f32_32x32x16_fp8_fp8 runs at the bf16
        // rate): v_mfma_scale_f32_32x32x64_f8f6f4 with both formats e4m3 and every E8M0 block scale = 2^0 (0x7f) is a
        // plain K = 64 fp8 contraction at twice the rate.  A lane's operand is 32 bytes = two of the 16-byte fragments it
        // already holds; A and B lanes pair the same fragments, so the k order is again a valid contraction order.  The
        // per-row fp32 scales stay in the epilogue (exactly the arithmetic of the two-instruction form: products of e4m3
        // values are exact in fp32 and the accumulation is fp32 either way; only the summation order inside K differs).
        typedef int i8v __attribute__((ext_vector_type(8)));
        typedef int i4v __attribute__((ext_vector_type(4)));
        auto cat = [](const bf16x8 lo, const bf16x8 hi) -> i8v {
          const i4v l = __builtin_bit_cast(i4v, lo), h = __builtin_bit_cast(i4v, hi);
          return i8v{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
        };
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          const i8v bq = cat(bf[2 * kp], bf[2 * kp + 1]);
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
            acc[mh * 2 + m2][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                bq, cat(af[m2][2 * kp], af[m2][2 * kp + 1]), acc[mh * 2 + m2][nt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
            acc[mh * 2 + m2][nt] = fma_frag(bf[ks], af[m2][ks], acc[mh * 2 + m2][nt]);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    auto enter_mfma = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto leave_mfma = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: K-tile 0 (A0, B1, A1, B0) and the first three half-tiles of K-tile 1
    stage_half(0, 0); stage_half(0, 3); stage_half(0, 1); stage_half(0, 2);
    if (nk > 1) {
      stage_half(1, 0); stage_half(1, 3); stage_half(1, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind group 0
    for (int t = 0; t < nk; ++t) {
      const char* As = smem + (t & 1) * STAGE_BYTES;
      const char* Bs = As + TILE_BYTES;
      // two-voxel convolution rows (conv_kw == 4): the first voxel's weights are zero at dx = 3, the second's at dx = 0,
      // so those k-tiles feed only one half of the N tile -- the other half's MFMA sections are skipped (uniform branch;
      // barriers, fragment reads and DMA keep their places, the phase just ends early)
      bool do_b0 = true, do_b1 = true;
      if constexpr (CONV) {
        if (p.conv_kw == 4 && p.N == 256) {  // the voxel halves coincide with the B half-tiles only at Cout = 128
          const int dx = (t >> (p.conv_cin_log2 - 6)) & 3;
          do_b0 = dx != 3, do_b1 = dx != 0;
        }
      }
      // p0: (A0, B0)
      load_a(As, 0);
      if (do_b0) load_b(Bs, 0, b0f);
      if (t + 1 < nk) stage_half(t + 1, 2);
      enter_mfma();
      if (do_b0) quad(0, 0, b0f);
      leave_mfma();
      // p1: (A0, B1)
      if (do_b1) load_b(Bs, 1, b1f);
      if (t + 2 < nk) stage_half(t + 2, 0);
      enter_mfma();
      if (do_b1) quad(0, 1, b1f);
      leave_mfma();
      // p2: (A1, B1)
      load_a(As, 1);
      if (t + 2 < nk) stage_half(t + 2, 3);
      enter_mfma();
      if (do_b1) quad(1, 1, b1f);
      leave_mfma();
      // p3: (A1, B0); the wait that publishes K-tile t+1
      if (t + 2 < nk) {
        stage_half(t + 2, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      enter_mfma();
      if (do_b0) quad(1, 0, b0f);
      leave_mfma();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  } else if constexpr (PIPE == 7) {
    // 4 waves x 128x128, ONE wave per SIMD, everything behind an MFMA.  Per K-tile a wave issues 64 MFMAs, 32 fragment
    // reads and 16 DMA instructions; nothing else can hide a stall, so each memory instruction sits in the issue shadow
    // of an MFMA (an MFMA holds the matrix pipe 32 cycles but the issue port 4):
    //     step 0: MFMA(ks0) + reads(ks1) + second half of the DMA of stage kt+1
    //     step 1: MFMA(ks1) + reads(ks2)        step 2: MFMA(ks2) + reads(ks3)
    //     vmcnt(0) lgkmcnt(0) barrier        (stage kt+1 published; every wave holds all fragments of stage kt)
    //     step 3: MFMA(ks3) + reads(ks0 of stage kt+1) + first half of the DMA of stage kt+2 into the freed buffer
    // The loop body is branch-free (first / steady / last-but-one / last iterations are separate copies) so the whole
    // K-tile is one scheduling region.
    constexpr int NDMA = 2 * LD_PER_OP;  // 16
    bf16x8 af0[4], bf0[NT], af1[4], bf1[NT];
    auto il_ds = [&]() {  // 8 DS reads behind the first 8 MFMAs, then the other 8 MFMAs
#pragma unroll
      for (int i = 0; i < 4 + NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - (4 + NT), 0);
    };
    auto il_ds_dma = [&]() {  // 8 DS reads behind MFMAs 0-7, 8 DMA issues behind MFMAs 8-15
#pragma unroll
      for (int i = 0; i < 4 + NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < NDMA / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
    auto iter = [&](int kt, auto dma0_c, auto bar_c, auto dma3_c) {
      constexpr bool DMA0 = decltype(dma0_c)::value, BAR = decltype(bar_c)::value, DMA3 = decltype(dma3_c)::value;
      const char* As = smem + (kt & 1) * STAGE_BYTES;
      const char* Bs = As + TILE_BYTES;
      load_frags(As, Bs, 1, af1, bf1);
      mma(af0, bf0);
      if (DMA0) {
#pragma unroll
        for (int w = NDMA / 2; w < NDMA; ++w) stage_piece((kt + 1) & 1, kt + 1, w);
        il_ds_dma();
      } else {
        il_ds();
      }
      __builtin_amdgcn_sched_barrier(0);
      load_frags(As, Bs, 2, af0, bf0);
      mma(af1, bf1);
      il_ds();
      __builtin_amdgcn_sched_barrier(0);
      load_frags(As, Bs, 3, af1, bf1);
      mma(af0, bf0);
      il_ds();
      __builtin_amdgcn_sched_barrier(0);
      if (BAR) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* An = smem + ((kt + 1) & 1) * STAGE_BYTES;
        load_frags(An, An + TILE_BYTES, 0, af0, bf0);
        mma(af1, bf1);
        if (DMA3) {
#pragma unroll
          for (int w = 0; w < NDMA / 2; ++w) stage_piece(kt & 1, kt + 2, w);
          il_ds_dma();
        } else {
          il_ds();
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        mma(af1, bf1);
      }
    };
    using T = IntC<1>;
    using F = IntC<0>;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) stage(1, 1);
    load_frags(smem, smem + TILE_BYTES, 0, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    if (nk == 1) {
      iter(0, F{}, F{}, F{});
    } else if (nk == 2) {
      iter(0, F{}, T{}, F{});
      iter(1, F{}, F{}, F{});
    } else {
      iter(0, F{}, T{}, T{});                                    // stage 1 went out whole in the prologue
      for (int kt = 1; kt + 2 < nk; ++kt) iter(kt, T{}, T{}, T{});
      iter(nk - 2, T{}, T{}, F{});
      iter(nk - 1, F{}, F{}, F{});
    }
  }

  if constexpr (PIPE == 8) {
    // 4 waves x 128x128, one wave per SIMD, a 4-stage ring of BK = 32 half-tiles.  Stage s of the ring is
    // [A: 256 rows x 64 B | B: 256 rows x 64 B] = 32 KiB; logical 16-byte slot q of row r sits at q ^ ((r >> 2) & 3) (16
    // consecutive rows x one slot cover all 64 banks; lanes l and l + 32 read slots q and q ^ 1 of one row).
    // Step j (one BK = 32 stage, 32 MFMAs from the fragments in F[j & 1]):
    //     s_waitcnt vmcnt(16) lgkmcnt(0)  my part of stage j+1 has landed (stages j+2, j+3 stay in flight), F[j & 1] is complete
    //     s_barrier                       => stage j+1 is whole, and every wave is done READING stage j (its fragments
    //                                        were fetched during step j-1), so that slot is free
    //     16 fragment reads of stage j+1 -> F[(j + 1) & 1], 8 DMA of stage j+4 -> the slot of stage j, 32 MFMAs:
    //     16 x {MFMA, ds_read}, 8 x {MFMA, DMA}, 8 MFMA.
    constexpr int SB = 32768, OPB = 16384;
    const int nst = p.K / 32;
    const int sw8 = (l31 >> 2) & 3;
    const elem_t* a8[4];
    const elem_t* b8[4];
    {
      const int r0 = tid >> 2, q = (tid & 3) ^ ((tid >> 4) & 3);   // row of round 0 and the LOGICAL slot this lane fetches
      const int am = (abl & 8) ? 0 : m0, bn = (abl & 8) ? 0 : n0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a8[i] = A + (int64_t)min(am + i * 64 + r0, p.M - 1) * p.lda + q * 8;
        b8[i] = B + (int64_t)min(bn + i * 64 + r0, p.N - 1) * p.ldb + q * 8;
      }
    }
    auto src_off = [&](int st) -> int64_t { return (int64_t)st * 32; };
    auto stage8 = [&](int st) {
      char* base = smem + (st & 3) * SB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(a8[i] + src_off(st)), (lptr_t)(base + (i * 256 + wave * 64) * 16), 16, 0, ALG_AUX_A);
        __builtin_amdgcn_global_load_lds((gptr_t)(b8[i] + src_off(st)), (lptr_t)(base + OPB + (i * 256 + wave * 64) * 16), 16, 0, ALG_AUX_B);
      }
    };
    const int a_off8 = (wm * 128 + l31) * 64, b_off8 = (wn * 128 + l31) * 64;
    auto load8 = [&](int st, bf16x8 (&af)[2][4], bf16x8 (&bfr)[2][4]) {
      const char* As = smem + (st & 3) * SB;
      const char* Bs = As + OPB;
#pragma unroll
      for (int ksl = 0; ksl < 2; ++ksl) {
        const int so = ((2 * ksl + h2) ^ sw8) * 16;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bfr[ksl][nt] = *(const bf16x8*)(Bs + b_off8 + nt * 32 * 64 + so);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[ksl][mt] = *(const bf16x8*)(As + a_off8 + mt * 32 * 64 + so);
      }
    };
    auto mma8 = [&](const bf16x8 (&af)[2][4], const bf16x8 (&bfr)[2][4]) {
#pragma unroll
      for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = fma_frag(bfr[ksl][nt], af[ksl][mt], acc[mt][nt]);
    };
    bf16x8 fa0[2][4], fb0[2][4], fa1[2][4], fb1[2][4];
    // One code copy for every step: past the end of K the DMA re-fetches the last stage into a slot nobody reads again and
    // the fragment reads fetch stale data into registers nobody uses (a few KiB of L2 traffic per tile; separate tail
    // copies made hipcc shuffle the 256 accumulators through scratch).
    auto step = [&](int j, bf16x8 (&ca)[2][4], bf16x8 (&cb)[2][4], bf16x8 (&na)[2][4], bf16x8 (&nb)[2][4]) {
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifndef ALG_P8_NO_READS   // build-time ablations (results are garbage; timing shows what the loop is bound by)
      load8(j + 1, na, nb);
#endif
#ifndef ALG_P8_NO_DMA
      {
        const int st = min(j + 4, nst - 1);
        char* base = smem + (j & 3) * SB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_global_load_lds((gptr_t)(a8[i] + src_off(st)), (lptr_t)(base + (i * 256 + wave * 64) * 16), 16, 0, ALG_AUX_A);
          __builtin_amdgcn_global_load_lds((gptr_t)(b8[i] + src_off(st)), (lptr_t)(base + OPB + (i * 256 + wave * 64) * 16), 16, 0, ALG_AUX_B);
        }
      }
#endif
      mma8(ca, cb);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    stage8(0);
    stage8(1);   // K is a multiple of 64: at least two stages
    stage8(min(2, nst - 1));
    stage8(min(3, nst - 1));
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load8(0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    for (int j = 0; j < nst; j += 2) {
      step(j, fa0, fb0, fa1, fb1);
      step(j + 1, fa1, fb1, fa0, fb0);
    }
    // the over-the-end DMA must have landed (and everybody must be past its last fragment read) before the epilogue
    // parks the tile in the ring
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: bias, activation, gate, residual, 8-byte stores ----
  // lane owns row m = m0 + wm*128 + mt*32 + l31 and, per register quad g, columns n = nbase + 8g + 4h2 + (0..3)
  const bf16_t* bias = (const bf16_t*)p.bias;
  const bf16_t* R = RES ? (const bf16_t*)p.R + (int64_t)b * p.strideR : nullptr;
  const bf16_t* gate = (RES && p.gate) ? (const bf16_t*)p.gate + (int64_t)b * p.strideGate : nullptr;
  const bool bias_row = p.flags & ALG_GEMM_BIAS_PER_ROW;
  const bool perm = p.flags & ALG_GEMM_PERMUTE_COLS;
  const int64_t gate_seg = (p.flags & ALG_GEMM_GATE_SEG_STRIDE) ? p.gate_seg_stride : p.N;
  const bool gate_f32 = RES && p.gate && (p.flags & ALG_GEMM_GATE_F32);  // Wan: fp32 gate, one rounding at the end
  bf16_t* Cb = (bf16_t*)p.C + (int64_t)b * p.strideC;
  const int ldc = (int)p.ldc, ldr = (int)p.ldr;
  // PERMUTE_COLS with a column origin (perm_col0: this GEMM fills columns [col0, col0 + N) of a wider permuted V^T row,
  // HunyuanVideo's [latents; text] joint sequence): the bit swap acts on the JOINT index, so the quad shortcut only
  // holds when col0 is a multiple of 16; otherwise take the element-wise path
  const int col0 = perm ? p.perm_col0 : 0;
  const bool n_vec = (p.N & 3) == 0 && (col0 & 15) == 0;  // whole quads are either inside or outside N
  // ---- coalesced epilogue of the ping-pong schedule: transpose through LDS ----
  // In the C^T accumulator layout a lane owns 4 consecutive columns of ONE row, so a wave's direct store instruction
  // touches 32 rows x 16 bytes: measured per-tile overhead 16 us (stores only) to 29 us (+ residual loads) against 77 us of
  // main loop at K = 3072.  After the K-loop the 128 KiB of LDS are free: every wave parks its 128x64 bf16 sub-tile
  // (after bias / scale / activation, exactly the bf16 value nn.Linear returns) in its own 16 KiB, reads it back with
  // 16 bytes per lane along rows, and applies gate / residual and stores with 16-byte accesses -- 64 contiguous bytes per
  // row segment instead of 16.  Bank-conflict-free both ways (chunk XOR (row >> 2) & 3).  Needs 16-byte aligned rows of
  // C / R / gate and N % 8 == 0; anything else takes the element-exact path below.
  // tile-local origin of the wave's 32x32 block (mt, nt) in either layout
  auto blk_row = [&](int mt) { return PP ? (mt >> 1) * 128 + wm * 64 + (mt & 1) * 32 : wm * 128 + mt * 32; };
  auto blk_col = [&](int nt) { return PP ? nt * 128 + wn * 32 : wn * (NT * 32) + nt * 32; };
  if constexpr (PP || PIPE == 8) {
    const bool staged = (p.N & 7) == 0 && (ldc & 7) == 0 && (p.strideC & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                        (col0 & 15) == 0 && !(abl & 32) &&
                        (!RES || ((ldr & 7) == 0 && (p.strideR & 7) == 0 && (((uintptr_t)p.R) & 15) == 0)) &&
                        (!(RES && p.gate) || ((p.strideGate & 7) == 0 && (gate_seg & 7) == 0 && (((uintptr_t)p.gate) & 15) == 0));
    if (staged) {
      char* const my = smem + wave * (4 * NT * 2048);   // 16 KiB (8 waves) or 32 KiB (4 waves) of the free ring
      // Epilogue operands first, ALL of them, before any arithmetic: the column bias (and fp8 column scales) of the wave's
      // NT x 4 quads are the same for its four row bands, the row bias / row scale is one value per band.  Loaded where they
      // were used (inside the quad loop, under `n < N`) every quad waited for its own L2 round trip: 32 serialised loads
      // per wave and tile.
      uint2 bcol[NT][4];
      float4 scol[FP8 ? NT : 1][FP8 ? 4 : 1];
      float brow4[4], a_sc4[4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + blk_col(nt) + 8 * g + 4 * h2;
          const int nc = n < p.N ? n : 0;                       // p.N is a multiple of 8 here: column 0 always exists
          bcol[nt][g] = (bias && !bias_row) ? *(const uint2*)(bias + nc) : make_uint2(0u, 0u);
          if constexpr (FP8) scol[nt][g] = *(const float4*)(p.b_scale + (int64_t)b * p.strideBScale + nc);
        }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = m0 + blk_row(mt) + l31;
        const int rowc = row < p.M ? row : p.M - 1;
        brow4[mt] = (bias && bias_row) ? bf2f(bias[rowc]) : 0.0f;
        a_sc4[mt] = FP8 ? p.a_scale[(int64_t)b * p.strideAScale + rowc] : 1.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
      // BR: per-row bias (the V^T projection) -- a workgroup-uniform choice, so the other form does not pay for an add of 0
      auto park_band = [&](auto mt_c, auto br_c) {
        constexpr int mt = decltype(mt_c)::value;
        constexpr bool BR = decltype(br_c)::value != 0;
        const float brow = brow4[mt];
        const float a_sc = a_sc4[mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float bv[4], sv[4] = {1.f, 1.f, 1.f, 1.f};
            unpack4(bcol[nt][g], bv);
            if constexpr (FP8) {
              const float4 s4 = scol[nt][g];
              sv[0] = s4.x * a_sc; sv[1] = s4.y * a_sc; sv[2] = s4.z * a_sc; sv[3] = s4.w * a_sc;
            }
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float dot = FP8 ? acc[mt][nt][4 * g + i] * sv[i] : acc[mt][nt][4 * g + i];
              // nn.Linear's bf16 result is the rounding pack_bf2 applies below; an activation works on that rounded value
              float x = BR ? dot + bv[i] + brow : dot + bv[i];
              if (ACT != ALG_ACT_NONE) x = act_apply(rbf(x), ACT);
              v[i] = x;
            }
            uint2 o;
            o.x = pack_bf2(v[0], v[1]);
            o.y = pack_bf2(v[2], v[3]);
            *(uint2*)(my + (mt * NT + nt) * 2048 + l31 * 64 + ((g ^ ((l31 >> 2) & 3)) * 16) + h2 * 8) = o;
          }
      };
      // (only the plain kernel is built in both forms: the per-row bias belongs to the V^T projection, which has neither
      // activation nor residual, and a second copy of those longer epilogues costs more in code size than the add)
      constexpr bool BR_SPLIT = !RES && ACT == ALG_ACT_NONE;
      if (!BR_SPLIT || (bias && bias_row)) {
        park_band(IntC<0>{}, IntC<1>{});
        park_band(IntC<1>{}, IntC<1>{});
        park_band(IntC<2>{}, IntC<1>{});
        park_band(IntC<3>{}, IntC<1>{});
      } else if constexpr (BR_SPLIT) {
        park_band(IntC<0>{}, IntC<0>{});
        park_band(IntC<1>{}, IntC<0>{});
        park_band(IntC<2>{}, IntC<0>{});
        park_band(IntC<3>{}, IntC<0>{});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave reads back only what it wrote itself
      // The residual may alias C (in-place update), so hipcc keeps every R load behind the previous iteration's store:
      // sixteen serialised HBM round trips per tile (measured: 21.7 us of per-tile overhead with a residual vs 6.9 us
      // without, scripts/gemm_k_sweep.py).  A thread reads exactly the elements it later writes, so all sixteen loads
      // can go out first; the accumulators are parked, their registers are free.  (Issuing them BEFORE the park, to hide
      // their latency under its arithmetic, needs 64 more live registers: 19 spills, and a kernel with scratch ran 10 %
      // slower -- measured.)
      constexpr int NIT = 8 * NT;   // 32x32 blocks x two 16-row halves
      uint4 rbuf[RES ? NIT : 1];
      if (RES) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int blk = it >> 1, mt = blk / NT, nt = blk % NT;
          const int rr = (it & 1) * 16 + (lane >> 2), ch = lane & 3;
          const int row = m0 + blk_row(mt) + rr;
          const int c0 = n0 + blk_col(nt) + ch * 8;
          rbuf[it] = (row < p.M && c0 < p.N) ? *(const uint4*)(R + row * ldr + c0) : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int blk = it >> 1, mt = blk / NT, nt = blk % NT;
        const int rr = (it & 1) * 16 + (lane >> 2), ch = lane & 3;
        const uint4 raw = *(const uint4*)(my + blk * 2048 + rr * 64 + ((ch ^ ((rr >> 2) & 3)) * 16));
        const int row = m0 + blk_row(mt) + rr;
        const int c0 = n0 + blk_col(nt) + ch * 8;
        if (row < p.M && c0 < p.N) {
          uint4 outv = raw;
          if (RES) {
            float x[8], r[8], gq[8];
            const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
            const uint4 rraw = rbuf[it];
            const uint32_t ru[4] = {rraw.x, rraw.y, rraw.z, rraw.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              x[2 * k] = __uint_as_float(u[k] << 16);
              x[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
              r[2 * k] = __uint_as_float(ru[k] << 16);
              r[2 * k + 1] = __uint_as_float(ru[k] & 0xffff0000u);
              gq[2 * k] = gq[2 * k + 1] = 1.0f;
            }
            const bool seg1 = row >= p.seg_split;
            if (gate) {
              if (gate_f32) {
                const float* gp = (const float*)p.gate + (int64_t)b * p.strideGate + (seg1 ? gate_seg : 0) + c0;
                const float4 g0 = *(const float4*)gp, g1 = *(const float4*)(gp + 4);
                gq[0] = g0.x; gq[1] = g0.y; gq[2] = g0.z; gq[3] = g0.w; gq[4] = g1.x; gq[5] = g1.y; gq[6] = g1.z; gq[7] = g1.w;
              } else {
                const uint4 graw = *(const uint4*)(gate + (seg1 ? gate_seg : 0) + c0);
                const uint32_t gu[4] = {graw.x, graw.y, graw.z, graw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  gq[2 * k] = __uint_as_float(gu[k] << 16);
                  gq[2 * k + 1] = __uint_as_float(gu[k] & 0xffff0000u);
                }
              }
            }
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = gate_f32 ? r[k] + gq[k] * x[k] : r[k] + rbf(gq[k] * x[k]);
            outv.x = pack_bf2(y[0], y[1]); outv.y = pack_bf2(y[2], y[3]);
            outv.z = pack_bf2(y[4], y[5]); outv.w = pack_bf2(y[6], y[7]);
          }
          bf16_t* crow = Cb + row * ldc;
          if (perm) {
            // columns c0..c0+3 (bit 2 = 0) and c0+4..c0+7 (bit 2 = 1) swap bit 2 with bit 3 = (c0 >> 3) & 1
            const int base = col0 + (c0 & ~12), b3 = (c0 >> 3) & 1;
            *(uint2*)(crow + base + (b3 << 2)) = make_uint2(outv.x, outv.y);
            *(uint2*)(crow + base + 8 + (b3 << 2)) = make_uint2(outv.z, outv.w);
          } else {
            *(uint4*)(crow + c0) = outv;
          }
        }
      }
      return;
    }
  }

  // one 32-row band per call with a compile-time index: with 256 accumulators hipcc stops fully unrolling a 4-deep mt
  // loop and the dynamically indexed accumulator array then lives in scratch (64 scratch stores per K-iteration)
  auto epilogue_band = [&](auto mt_c) {
    constexpr int mt = decltype(mt_c)::value;
    const int row = PP ? m0 + (mt >> 1) * 128 + wm * 64 + (mt & 1) * 32 + l31 : m0 + wm * 128 + mt * 32 + l31;
    const bool row_ok = row < p.M;
    const int rowc = row_ok ? row : p.M - 1;
    const float brow = (bias && bias_row) ? bf2f(bias[rowc]) : 0.0f;
    const bool seg1 = row >= p.seg_split;
    const float a_sc = FP8 ? p.a_scale[(int64_t)b * p.strideAScale + rowc] : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = PP ? n0 + nt * 128 + wn * 32 + 8 * g + 4 * h2 : n0 + wn * (NT * 32) + nt * 32 + 8 * g + 4 * h2;
        if (n_vec) {
          if (n < p.N) {
            float bv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias && !bias_row) unpack4(*(const uint2*)(bias + n), bv);
            if (RES && gate) {
              if (gate_f32) {
                const float4 g4 = *(const float4*)((const float*)p.gate + (int64_t)b * p.strideGate + (seg1 ? gate_seg : 0) + n);
                gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
              } else {
                unpack4(*(const uint2*)(gate + (seg1 ? gate_seg : 0) + n), gv);
              }
            }
            if (RES) unpack4(*(const uint2*)(R + rowc * ldr + n), rv);
            float sv[4] = {1.f, 1.f, 1.f, 1.f};
            if (FP8) {
              const float4 s4 = *(const float4*)(p.b_scale + (int64_t)b * p.strideBScale + n);
              sv[0] = s4.x * a_sc; sv[1] = s4.y * a_sc; sv[2] = s4.z * a_sc; sv[3] = s4.w * a_sc;
            }
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float dot = FP8 ? acc[mt][nt][4 * g + i] * sv[i] : acc[mt][nt][4 * g + i];
              float x = rbf(dot + bv[i] + brow);  // nn.Linear returns a bf16 tensor
              if (ACT != ALG_ACT_NONE) x = rbf(act_apply(x, ACT));
              if (RES) x = gate_f32 ? rv[i] + gv[i] * x : rv[i] + rbf(gv[i] * x);
              v[i] = x;
            }
            if (row_ok) {
              // PERMUTE_COLS swaps index bits 2 and 3: quad (g, h2) lands where (h2, g & 1) says
              const int nc = perm ? col0 + ((n & ~12) | (h2 << 3) | ((g & 1) << 2)) : n;
              uint2 o;
              o.x = pack_bf2(v[0], v[1]);
              o.y = pack_bf2(v[2], v[3]);
              *(uint2*)(Cb + row * ldc + nc) = o;
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int nn = n + i;
            if (nn < p.N && row_ok) {
              const float dot = FP8 ? acc[mt][nt][4 * g + i] * (a_sc * p.b_scale[(int64_t)b * p.strideBScale + nn])
                                    : acc[mt][nt][4 * g + i];
              float x = rbf(dot + ((bias && !bias_row) ? bf2f(bias[nn]) : 0.0f) + brow);
              if (ACT != ALG_ACT_NONE) x = rbf(act_apply(x, ACT));
              if (RES) {
                if (gate_f32) {
                  x = bf2f(R[rowc * ldr + nn]) +
                      ((const float*)p.gate)[(int64_t)b * p.strideGate + (seg1 ? gate_seg : 0) + nn] * x;
                } else {
                  const float gg = gate ? bf2f(gate[(seg1 ? gate_seg : 0) + nn]) : 1.0f;
                  x = bf2f(R[rowc * ldr + nn]) + rbf(gg * x);
                }
              }
              const int nj = nn + col0;
              const int nc = perm ? ((nj & ~12) | ((nj & 4) << 1) | ((nj & 8) >> 1)) : nn;
              Cb[row * ldc + nc] = f2bf(x);
            }
          }
        }
      }
    }
  };
#ifndef ALG_EPI_STAGED_ONLY   // analysis builds (instruction counts of the staged epilogue alone) leave the fallback out
  epilogue_band(IntC<0>{});
  epilogue_band(IntC<1>{});
  epilogue_band(IntC<2>{});
  epilogue_band(IntC<3>{});
#endif
  };  // do_tile

  // ---- logical workgroup -> (batch, m_tile, n_tile): XCD-contiguous, grouped along M ----
  // Logical index L keeps its XCD (gridDim.x is a multiple of 8 when the launch is persistent), so the per-XCD tile
  // order -- and with it the L2 reuse -- is the one a full grid would have.
  const int total_wg = tiles * p.batch;
  for (int L = blockIdx.x; L < total_wg; L += gridDim.x) {
    const int xcd = L & 7, q = total_wg >> 3, r = total_wg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    if (L != (int)blockIdx.x) __syncthreads();  // the previous tile's epilogue is done with the LDS ring
    do_tile(wg);
  }
}

inline int gemm_pipe() {
  const char* e = getenv("ALG_GEMM_PIPE");
  const int v = e ? atoi(e) : 6;  // default: 8-wave ping-pong over half-tiles (fastest measured)
  return (v == 0 || v == 6 || v == 7 || v == 8) ? v : 6;
}

// persistent launch: one workgroup per CU walks the tile list (saves a workgroup launch + teardown per tile: at K = 3072
// a tile's main loop is only ~80 us).  ALG_GEMM_PERSIST=0 launches one workgroup per tile as before.
inline unsigned gemm_grid(int64_t nwg) {
  static std::atomic<int> cus{0};
  int n = cus.load();
  if (n == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    n &= ~7;
    if (n < 8) n = 8;
    cus.store(n);
  }
  const char* e = getenv("ALG_GEMM_PERSIST");
  if (e && e[0] == '0') return (unsigned)nwg;
  return (unsigned)(nwg < n ? nwg : n);
}

inline int gemm_group_m() {
  const char* e = getenv("ALG_GEMM_GROUP_M");
  const int v = e ? atoi(e) : GROUP_M;
  const char* a = getenv("ALG_GEMM_ABLATE");
  return ((v < 1 || v > 0xffff) ? GROUP_M : v) | ((a ? atoi(a) : 0) << 16);
}

}  // namespace alg

using namespace alg;

template <int PIPE, int WNW = 4, bool FP8 = false>
int launch_gemm(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s) {
  static std::atomic<bool> attr_set{false};  // idempotent one-time setup; racing first calls both succeed
  if (!attr_set) {
    const void* fns[4] = {(const void*)gemm_bf16_kernel<ALG_ACT_NONE, true, PIPE, WNW, FP8>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_GELU_TANH, false, PIPE, WNW, FP8>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_SILU, false, PIPE, WNW, FP8>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_NONE, false, PIPE, WNW, FP8>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
      if (e != hipSuccess) {
        set_error("alg_gemm_bf16: hipFuncSetAttribute(%d B LDS): %s", GEMM_LDS, hipGetErrorString(e));
        return ALG_ELAUNCH;
      }
    }
    attr_set = true;
  }
  const dim3 grid(gemm_grid(nwg)), block(2 * WNW * 64);
  const int gm = gemm_group_m();
  if (a->R) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, true, PIPE, WNW, FP8>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm);
  } else if (a->act == ALG_ACT_GELU_TANH) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_GELU_TANH, false, PIPE, WNW, FP8>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm);
  } else if (a->act == ALG_ACT_SILU) {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_SILU, false, PIPE, WNW, FP8>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm);
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, false, PIPE, WNW, FP8>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm);
  }
  return check_launch("alg_gemm_bf16");
}

