// Flash attention forward, head_dim 64, the long self-attention form of the CogVideoX DiT: 64 queries per wave, one wave per
// SIMD, the steady-state KV loop ONE generated asm statement (round 5) -- the d = 64 sibling of attention128_q64.hip (read its
// header for the construction; this file states what differs).
//
//   * Main launch of the PRE-SCALED call only (ALG_ATTN_Q_PRESCALED: Q carries scale * log2(e), scores arrive in log2 units) with
//     the running offset snapped to ZERO on the first tile (attention.hip, softmax_tile_zero): p = exp2(s), no subtraction.  The
//     statement (attn64_q64_loop.inc, scripts/gen_attn_q64.py, Cfg(64, fma=False)) is entered only by waves whose offsets are all
//     zero; everything else -- tile 0, the tail, refused tiles, rows with a non-zero offset -- runs the C++ tile body below.
//   * Per 64-key tile and wave 32 MFMAs (16 PV(t-1) + 16 QK(t+1)) against the same 64 x 64 scores as at d = 128: one score PAIR
//     per MFMA gap (2 exp2, 1 cvt_pk, 2 adds) + half a fragment read.  The 8-wave 32-query statement (attention.hip) issues a
//     whole fragment read per MFMA and stages nothing less; what this form buys is half the LDS instructions per MFMA.
//   * K and V^T tiles are both 64 rows x 128 bytes (8 KiB, swizzle (row >> 1) & 7): two 4-slot rings = 64 KiB of LDS; four DMA
//     pieces per wave and tile (counted wait vmcnt(4)).
//   * The split-KV tail of alg_flash_attn_d64 stays on attention.hip's kernel: this one replaces the MAIN launch (same 256-query
//     units, same block -> (head, q block) order).
// Selected by ALG_ATTN_PP=6 (see alg_hip.h for the default).
#include <stdlib.h>

#include "common.h"
#include "attn64_q64_loop.inc"

namespace alg {
namespace a64q {

constexpr int NW = 4;
constexpr int QW = 64;
constexpr int KVB = 64;
constexpr int TILE = KVB * 64 * 2;       // 8 KiB: K tile = V^T tile
constexpr int LDS_BYTES = 8 * TILE;
constexpr int MIN_TILES = 12;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct P {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int batch, heads, S, q_blocks;
  int64_t q_bs, q_rs, vt_bs, vt_rs, o_bs, o_rs;
  uint64_t* clk;
  int clk_slots;
};

__global__ __launch_bounds__(NW * 64) void flash_attn_d64_q64_kernel(const P p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* const k_ring = smem;
  char* const v_ring = smem + 4 * TILE;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nbh = p.batch * p.heads;
  int bh, qb;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int slot = idx / p.q_blocks;
    qb = idx - slot * p.q_blocks;
    bh = slot * 8 + xcd;
    if (bh >= nbh) return;
  }
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int S = p.S;
  const bool tap = p.clk != nullptr && (blockIdx.x & 63) == 0 && (int)(blockIdx.x >> 6) < p.clk_slots && wave == 0;   // clock tap: see attention.hip
  uint64_t tap_c0 = 0, tap_r0 = 0;
  if (tap) {
    tap_c0 = __builtin_readcyclecounter();
    tap_r0 = wall_clock64();
  }
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* K = p.k + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 64 * p.vt_rs;
  const int T = (S + KVB - 1) / KVB;
  const bool ragged = (S & (KVB - 1)) != 0;
  // O^T of the wave's two query halves: tile (qh, dt) = oa[2 qh + dt]
  f32x16 oa[4];
#pragma unroll
  for (int i = 0; i < 64; ++i) oa[i >> 4][i & 15] = 0.0f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};

  struct LaneCtx {
    int l31, h2, tid, srow, sslot, q_row, row_off, sw;
  };
  auto make_ctx = [&](int lane) -> LaneCtx {
    LaneCtx x;
    x.l31 = lane & 31, x.h2 = lane >> 5, x.tid = wave * 64 + lane;
    x.srow = x.tid >> 3, x.sslot = (x.tid & 7) ^ ((x.tid >> 4) & 7);          // + 32 rows per DMA piece (K and V^T alike)
    x.q_row = qb * (NW * QW) + wave * QW + x.l31;                              // query of half 0; half 1: + 32
    x.row_off = x.l31 * 128, x.sw = (x.l31 >> 1) & 7;
    return x;
  };
  auto fresh_lane = [&]() -> int {
    int z = 0;
    asm volatile("" : "+s"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  };
  auto stage_k = [&](const LaneCtx& x, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* ks = K + (int64_t)min(t * KVB + x.srow + 32 * i, S - 1) * p.q_rs + x.sslot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + (t & 3) * TILE + (i * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](const LaneCtx& x, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(VT + (int64_t)(x.srow + 32 * i) * p.vt_rs + x.sslot * 8 + min(t, T - 1) * KVB),
                                       (lptr_t)(v_ring + (t & 3) * TILE + (i * 4 + wave) * 1024), 16, 0, 0);
  };
  // ONE tile in the straight form (builtin MFMAs on C++ values: hipcc sees every hazard): protocol, S = K Q^T, the zero-offset
  // lazy softmax of attention.hip's softmax_tile_zero with fp32 row sums, O += V^T P^T -- for both query halves.
  auto straight_tile = [&](const LaneCtx& x, int t, bool top_done) {
    if (!top_done) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the previous iteration's four DMAs
      __syncthreads();
      stage_k(x, t + 3);   // (past the end: clamped sources; the DMA count per iteration must not depend on t)
      stage_v(x, t + 2);
    }
    const char* Ks = k_ring + (t & 3) * TILE + x.row_off;
    const char* Vs = v_ring + (t & 3) * TILE + x.row_off;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      bf16x8 qf[4];
      const bf16_t* qp = Q + (int64_t)min(x.q_row + 32 * qh, S - 1) * p.q_rs + x.h2 * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
      f32x16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const bf16x8 kf = *(const bf16x8*)(Ks + sub * 4096 + (((2 * ks + x.h2) ^ x.sw) * 16));
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[sub], 0, 0, 0);
        }
      if (ragged && t == T - 1) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kv = t * KVB + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * x.h2;
            if (kv >= S) s[sub][e] = -INFINITY;
          }
      }
      bf16x8 pf[4];
      auto probs = [&](float m) -> float {
        float psum = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p0 = __builtin_amdgcn_exp2f(s[sub][8 * g + 2 * j] - m);
              const float p1 = __builtin_amdgcn_exp2f(s[sub][8 * g + 2 * j + 1] - m);
              pk.u[j] = pack_bf2(p0, p1);
              psum += p0 + p1;       // fp32 sums of the unrounded probabilities, as inside the statement
            }
            pf[sub * 2 + g] = pk.v;
          }
        return psum;
      };
      float psum = probs(m_run[qh]);   // (x - 0 = x: the zero-offset rows compute exactly what the statement computes)
      if (__any(!(psum < ALG_LAZY_SUM_LIMIT))) {  // 2^80; also inf (first tile: m = -inf) and NaN
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        float m_new = fmaxf(m_run[qh], mt);
        if (m_run[qh] == -INFINITY && fabsf(m_new) < 64.0f) m_new = 0.0f;  // first tile: snap the offset to zero when it is safe
        const float alpha = __builtin_amdgcn_exp2f(m_run[qh] - m_new);
        m_run[qh] = m_new;
        l_run[qh] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) oa[2 * qh + dt] *= alpha;
        psum = probs(m_run[qh]);
      }
      l_run[qh] += psum;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 vf = *(const bf16x8*)(Vs + dt * 4096 + (((2 * kk + x.h2) ^ x.sw) * 16));
          oa[2 * qh + dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], oa[2 * qh + dt], 0, 0, 0);
        }
    }
  };

  // the statement runs iterations t < tend: QK(t + 1) must not touch the masked (ragged) last tile; DMA pieces past the end of a
  // panel fetch nothing (the buffer descriptors' num_records)
  const int tend = ragged ? T - 2 : T - 1;
  int t = 1;
  bool top_done = false;
  {
    const LaneCtx x = make_ctx(fresh_lane());
    stage_k(x, 0);
    stage_k(x, 1);
    stage_v(x, 0);
    stage_v(x, 0);       // (filler: four DMAs per batch)
    stage_k(x, 2);       // the batch "iteration -1" would have issued: K(2), V(1)
    stage_v(x, 1);
    straight_tile(x, 0, false);     // tile 0: establishes the running offset (snapped to zero when its scores allow)
  }
  for (;;) {
    if ((t & 3) == 1 && t < tend && __all(m_run[0] == 0.0f && m_run[1] == 0.0f)) {
      const LaneCtx x = make_ctx(fresh_lane());
      auto sreg = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
      auto uniform64 = [](const void* ptr) -> uint64_t {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
      };
      const uint32_t kl = (uint32_t)(uintptr_t)(lptr_t)k_ring, vl = (uint32_t)(uintptr_t)(lptr_t)v_ring;
      int lk[4], lv[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int fl = x.row_off + (((2 * ks + x.h2) ^ x.sw) * 16);
        lk[ks] = kl + fl, lv[ks] = vl + fl;
      }
      // DMA: the lane's byte offset inside a tile (constant) against raw buffer descriptors whose base is the tile the statement
      // fetches first -- K(t + 3), V^T(t + 2) -- and whose num_records is what is left of this (batch, head) panel from there
      // (0 once the tile lies past the end: such pieces fetch nothing)
      int kvo[2], vvo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kvo[i] = (int)(((int64_t)(x.srow + 32 * i) * p.q_rs + x.sslot * 8) * 2);
        vvo[i] = (int)(((int64_t)(x.srow + 32 * i) * p.vt_rs + x.sslot * 8) * 2);
      }
      const int64_t k_tile_bytes = (int64_t)KVB * p.q_rs * 2;
      const int64_t k_left = ((int64_t)(S - 1) * p.q_rs + 64) * 2 - (int64_t)(t + 3) * k_tile_bytes;
      const int64_t v_left = (int64_t)64 * p.vt_rs * 2 - (int64_t)(t + 2) * KVB * 2;
      const uint64_t kt = uniform64((const char*)K + (int64_t)(t + 3) * k_tile_bytes);
      const uint64_t vtb = uniform64((const char*)VT + (int64_t)(t + 2) * KVB * 2);
      const int kd0 = sreg((int)(uint32_t)kt), kd1 = sreg((int)(uint32_t)(kt >> 32) & 0xffff), kd2 = sreg((int)(uint32_t)(k_left > 0 ? k_left : 0));
      const int vd0 = sreg((int)(uint32_t)vtb), vd1 = sreg((int)(uint32_t)(vtb >> 32) & 0xffff), vd2 = sreg((int)(uint32_t)(v_left > 0 ? v_left : 0));
      const int d3 = sreg(0x00020000);
      const int qvo0 = (int)(((int64_t)min(x.q_row, S - 1) * p.q_rs + x.h2 * 8) * 2);
      const int qvo1 = (int)(((int64_t)min(x.q_row + 32, S - 1) * p.q_rs + x.h2 * 8) * 2);
      const uint64_t qbs = uniform64(Q);
      const int kstep = sreg((int)k_tile_bytes), tend_s = sreg(tend);
      const int wk = sreg((int)kl + wave * 1024), wv = sreg((int)vl + wave * 1024);
      int ts = sreg(t), code;
      asm volatile(ALG_ATTN64_Q64_LOOP_ASM
                   : ALG_ATTN64_Q64_O_OPERANDS(oa), [l0] "+v"(l_run[0]), [l1] "+v"(l_run[1]), [t] "+s"(ts), [code] "=&s"(code)
                   : [lk0] "v"(lk[0]), [lk1] "v"(lk[1]), [lk2] "v"(lk[2]), [lk3] "v"(lk[3]), [lv0] "v"(lv[0]), [lv1] "v"(lv[1]),
                     [lv2] "v"(lv[2]), [lv3] "v"(lv[3]), [kvo0] "v"(kvo[0]), [kvo1] "v"(kvo[1]), [vvo0] "v"(vvo[0]), [vvo1] "v"(vvo[1]),
                     [qvo0] "v"(qvo0), [qvo1] "v"(qvo1), [kd0] "s"(kd0), [kd1] "s"(kd1), [kd2] "s"(kd2), [kd3] "s"(d3),
                     [vd0] "s"(vd0), [vd1] "s"(vd1), [vd2] "s"(vd2), [vd3] "s"(d3),
                     [qb] "s"(qbs), [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                   : "memory", "vcc", "scc", ALG_ATTN64_Q64_CLOBBERS);
      t = ts;
      top_done = code != 0;   // 1: iteration t's protocol is done, softmax(t) is not: tile t is redone below
    }
    if (t >= T) break;
    const LaneCtx x = make_ctx(fresh_lane());
    straight_tile(x, t, top_done);   // a tile behind the statement, or one it refused (then back into it at the next t = 1 mod 4)
    top_done = false;
    ++t;
  }

  const LaneCtx x = make_ctx(fresh_lane());
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const float l_tot = l_run[qh] + __shfl_xor(l_run[qh], 32, 64);
    const float inv = 1.0f / l_tot;
    const int q_row = x.q_row + 32 * qh;
    if (q_row < S) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * x.h2;
          uint2 v;
          v.x = pack_bf2(oa[2 * qh + dt][4 * g] * inv, oa[2 * qh + dt][4 * g + 1] * inv);
          v.y = pack_bf2(oa[2 * qh + dt][4 * g + 2] * inv, oa[2 * qh + dt][4 * g + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
  if (tap && x.l31 == 0 && x.h2 == 0) {
    uint64_t* cp = p.clk + (size_t)(blockIdx.x >> 6) * 4;   // one workgroup owns a slot (block / 64 < slots)
    cp[0] = tap_c0, cp[1] = tap_r0, cp[2] = __builtin_readcyclecounter(), cp[3] = wall_clock64();
  }
}

}  // namespace a64q

// Main launch of the pre-scaled call on `blocks` workgroups.  Returns ALG_OK when launched, 1 when this call is not covered (the
// caller launches attention.hip's main kernel instead), < 0 on error.
int flash_attn_d64_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S, int q_blocks,
                       int64_t q_bs, int64_t q_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs, int64_t o_rs, unsigned blocks,
                       hipStream_t stream) {
  using namespace a64q;
  if (opt(OPT_ATTN_PP) != 6 || (S + KVB - 1) / KVB < MIN_TILES || blocks == 0) return 1;
  if (q_blocks != (S + NW * QW - 1) / (NW * QW)) return 1;
  if ((int64_t)(S + 4 * KVB) * q_rs * 2 >= (1ll << 31) || (int64_t)65 * vt_rs * 2 >= (1ll << 31)) return 1;   // 31-bit byte offsets
  if (vt_rs < (int64_t)((S + KVB - 1) / KVB) * KVB) return 1;
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.S = S; p.q_blocks = q_blocks;
  p.q_bs = q_bs; p.q_rs = q_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.clk = clock_tap_for((hipStream_t)stream, &p.clk_slots);
  hipLaunchKernelGGL(flash_attn_d64_q64_kernel, dim3(blocks), dim3(NW * 64), 0, stream, p);
  return check_launch("alg_flash_attn_d64");
}

}  // namespace alg
