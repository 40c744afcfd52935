// Flash attention forward, head_dim 128, the LONG self-attention form of the Wan / HunyuanVideo DiTs: 64 queries per wave, one
// wave per SIMD, the steady-state KV loop ONE generated asm statement (round 5).
//
// Why 64 queries per wave: every K / V^T fragment read from LDS feeds TWO MFMAs (the wave's two 32-query halves) and a 64-key
// tile is staged once per 256 queries with four waves -- half the LDS instructions and half the L2 -> LDS traffic per FLOP of the
// 32-query kernels (attention128_pipe.hip, attention128.hip).  Round 4 shipped that idea with the MFMAs and fragment reads as
// separate inline-asm statements and everything between them scheduled by hipcc: 1280-1320 TFLOP/s at 62 % matrix-pipe busy, and
// a three-round hunt for a register hipcc recycled in the shadow of an MFMA it cannot see (profiles/r4_attention128_q64_probe.txt).
// Round 5 replaces that body by the construction of the 32-query statements: scripts/gen_attn_q64.py emits the whole steady
// state -- per 64-key tile and wave 64 MFMAs (PV(t-1) and QK(t+1)), 32 fragment reads, the 8 DMA pieces of the wave and the 224
// VALU instructions of softmax(t), each at a fixed place between two MFMAs -- as attn128_q64_loop.inc, and NO compiler-scheduled
// instruction sits between the first and the last MFMA of the statement (the hazard class of round 4 cannot occur: the statement
// ends behind s_nops that cover its last MFMA).  The generated text is checked on the CPU by an instruction-level emulator under
// the weakest memory ordering the ISA allows (tests/test_attn_q64_statement_cpu.py).
//
// This file is the frame, and everything in it is plain C++ the compiler sees whole: workgroup -> (head, q block) order, operand
// layouts and swizzles of attention128_pipe.hip, O as eight f32x16 values (handed to the statement as "+a" operands), a C++
// tile body for tile 0 (where the lazy running max is established), the last one or two tiles (the masked one) and any tile the statement
// refuses (row sum outside [0, 2^80): exact max / rescale), all under the statement's collective protocol
//     top of iteration t:  s_waitcnt vmcnt(8); s_barrier; DMA K(t+3) -> K slot (t+3) & 3, V^T(t+2) -> V slot (t+2) & 3
// so the waves of a workgroup may be inside or outside the statement independently.  A wave that left the statement RE-ENTERS
// it at the next t = 1 (mod 4) (round 4's single-statement kernels stayed outside for good).
//
// DEFAULT for non-causal, ungrouped attention over at least POLICY_TILES KV tiles; ALG_ATTN128_Q64=0 switches it off (the
// 32-query pipelined kernel takes over), =2 takes every call of at least MIN_TILES tiles (tests).
#include <stdlib.h>

#include "common.h"
#include "attn128_q64_loop.inc"

namespace alg {
namespace a128q {

constexpr int NW = 4;
constexpr int QW = 64;                   // queries per wave
constexpr int KVB = 64;
constexpr int TILE = 16384;              // K tile = V^T tile
constexpr int LDS_BYTES = 8 * TILE;      // 4 K slots + 4 V^T slots
constexpr int MIN_TILES = 8;             // what the kernel can take (ALG_ATTN128_Q64=2)
constexpr int POLICY_TILES = 64;         // what it takes by default: 4,096 keys and more

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct P {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int batch, heads, Sq, Skv, q_blocks;
  int64_t q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs;
  float scale_log2;
  uint64_t* clk;   // clock tap (calibrate.hip: alg_attn_clock_tap) or NULL
  int clk_slots;
  int use_statement;   // 0: every tile through the C++ tile body (ALG_ATTN128_Q64=3: tests of the frame on its own)
};

__global__ __launch_bounds__(NW * 64) void flash_attn_d128_q64_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int Sq = p.Sq, Skv = p.Skv;
  const bool tap = p.clk != nullptr && (blockIdx.x & 63) == 0 && (int)(blockIdx.x >> 6) < p.clk_slots && wave == 0;   // clock tap: see attention.hip
  uint64_t tap_c0 = 0, tap_r0 = 0;
  if (tap) {
    tap_c0 = __builtin_readcyclecounter();
    tap_r0 = wall_clock64();
  }
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 128;
  const bf16_t* K = p.k + (int64_t)b * p.k_bs + h * 128;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 128 * p.vt_rs;
  const int T = (Skv + KVB - 1) / KVB;
  const bool ragged = (Skv & (KVB - 1)) != 0;
  const float c = p.scale_log2;
  // O^T of the wave's two query halves: tile (qh, dt) = oa[4 qh + dt], lane (q = l31, h2) register e <-> d = 32 dt + (e & 3) + 8 (e >> 2) + 4 h2
  f32x16 oa[8];
#pragma unroll
  for (int i = 0; i < 128; ++i) oa[i >> 4][i & 15] = 0.0f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};

  // everything lane-derived is rebuilt from a freshly laundered lane id by each phase (in front of, inside the operand set-up of,
  // and behind the statement): none of it is live across the statement, whose literal registers leave the compiler v[0:51]
  struct LaneCtx {
    int l31, h2, tid, k_row, k_slot, v_row, v_slot, q_row, k_row_off, k_sw, v_row_off, v_sw;
  };
  auto make_ctx = [&](int lane) -> LaneCtx {
    LaneCtx x;
    x.l31 = lane & 31, x.h2 = lane >> 5, x.tid = wave * 64 + lane;
    x.k_row = x.tid >> 4, x.k_slot = (x.tid & 15) ^ ((x.tid >> 4) & 15);      // + 16 rows per DMA piece
    x.v_row = x.tid >> 3, x.v_slot = (x.tid & 7) ^ ((x.tid >> 4) & 7);        // + 32 d-rows per DMA piece
    x.q_row = qb * (NW * QW) + wave * QW + x.l31;                              // query of half 0; half 1: + 32
    x.k_row_off = x.l31 * 256, x.k_sw = x.l31 & 15;
    x.v_row_off = x.l31 * 128, x.v_sw = (x.l31 >> 1) & 7;
    return x;
  };
  auto fresh_lane = [&]() -> int {
    int z = 0;
    asm volatile("" : "+s"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  };
  auto stage_k = [&](const LaneCtx& x, int t) {
    const int kv0 = min(t, T - 1) * KVB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* ks = K + (int64_t)min(kv0 + x.k_row + 16 * i, Skv - 1) * p.k_rs + x.k_slot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + (t & 3) * TILE + (i * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](const LaneCtx& x, int t) {
    const int kv0 = min(t, T - 1) * KVB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(VT + (int64_t)(x.v_row + 32 * i) * p.vt_rs + x.v_slot * 8 + kv0),
                                       (lptr_t)(v_ring + (t & 3) * TILE + (i * 4 + wave) * 1024), 16, 0, 0);
  };
  // ONE tile in the straight form: protocol (unless done), S = K Q^T for both query halves, lazy softmax with the exact path in
  // line, O += V^T P^T.  Builtin MFMAs on C++ values: hipcc sees every hazard.  Runs tile 0, the tail and refused tiles only.
  auto straight_tile = [&](const LaneCtx& x, int t, bool top_done) {
    if (!top_done) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but the previous iteration's eight DMAs
      __syncthreads();
      stage_k(x, t + 3);   // (past the end: clamped sources; the DMA count per iteration must not depend on t)
      stage_v(x, t + 2);
    }
    const char* Ks = k_ring + (t & 3) * TILE + x.k_row_off;
    const char* Vs = v_ring + (t & 3) * TILE + x.v_row_off;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      bf16x8 qf[8];
      const bf16_t* qp = Q + (int64_t)min(x.q_row + 32 * qh, Sq - 1) * p.q_rs + x.h2 * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
      f32x16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const bf16x8 kf = *(const bf16x8*)(Ks + sub * 8192 + (((2 * ks + x.h2) ^ x.k_sw) * 16));
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[sub], 0, 0, 0);
        }
      if (ragged && t == T - 1) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kv = t * KVB + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * x.h2;
            if (kv >= Skv) s[sub][e] = -INFINITY;
          }
      }
      bf16x8 pf[4];
      auto probs = [&](float mc) -> float {
        float psum = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p0 = __builtin_amdgcn_exp2f(s[sub][8 * g + 2 * j] * c - mc);
              const float p1 = __builtin_amdgcn_exp2f(s[sub][8 * g + 2 * j + 1] * c - mc);
              pk.u[j] = pack_bf2(p0, p1);
              psum += p0 + p1;       // fp32 sums of the unrounded probabilities, as inside the statement
            }
            pf[sub * 2 + g] = pk.v;
          }
        return psum;
      };
      float psum = probs(m_run[qh] * c);
      if (__any(!(psum < ALG_LAZY_SUM_LIMIT))) {  // 2^80; also inf / NaN (tile 0: m = -inf)
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
        {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
          mt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        const float m_new = fmaxf(m_run[qh], mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qh] - m_new) * c);
        m_run[qh] = m_new;
        l_run[qh] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oa[4 * qh + dt] *= alpha;
        psum = probs(m_run[qh] * c);
      }
      l_run[qh] += psum;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 vf = *(const bf16x8*)(Vs + dt * 4096 + (((2 * kk + x.h2) ^ x.v_sw) * 16));
          oa[4 * qh + dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], oa[4 * qh + dt], 0, 0, 0);
        }
    }
  };

  // the statement runs iterations t < tend: QK(t + 1) must not touch the masked (ragged) last tile.  Its DMA of K(t + 3) / V^T(t + 2)
  // reaches past the end in the last iterations: the buffer descriptors' num_records end the panel, such pieces fetch nothing
  const int tend = ragged ? T - 2 : T - 1;
  int t = 1;
  bool top_done = false;
  {
    const LaneCtx x = make_ctx(fresh_lane());
    stage_k(x, 0);
    stage_k(x, 1);
    stage_v(x, 0);
    stage_v(x, 0);       // (filler: eight DMAs per batch)
    stage_k(x, 2);       // the batch "iteration -1" would have issued: K(2), V(1)
    stage_v(x, 1);
    straight_tile(x, 0, false);     // tile 0: establishes the running max of both query halves
  }
  for (;;) {
    if (p.use_statement && (t & 3) == 1 && t < tend && __all(m_run[0] > -3.0e38f && m_run[1] > -3.0e38f)) {
      const LaneCtx x = make_ctx(fresh_lane());
      auto sreg = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
      auto uniform64 = [](const void* ptr) -> uint64_t {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
      };
      const uint32_t kl = (uint32_t)(uintptr_t)(lptr_t)k_ring, vl = (uint32_t)(uintptr_t)(lptr_t)v_ring;
      int lk[8], lv[4];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) lk[ks] = kl + x.k_row_off + (((2 * ks + x.h2) ^ x.k_sw) * 16);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) lv[kk] = vl + x.v_row_off + (((2 * kk + x.h2) ^ x.v_sw) * 16);
      // DMA: the lane's byte offset inside a tile (constant) against raw buffer descriptors whose base is the tile the statement
      // fetches first -- K(t + 3), V^T(t + 2) -- and whose num_records is what is left of this (batch, head) panel from there
      // (0 once the tile lies past the end: the statement prefetches past the last tile it computes, and such pieces fetch nothing)
      int kvo[4], vvo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kvo[i] = (int)(((int64_t)(x.k_row + 16 * i) * p.k_rs + x.k_slot * 8) * 2);
        vvo[i] = (int)(((int64_t)(x.v_row + 32 * i) * p.vt_rs + x.v_slot * 8) * 2);
      }
      const int64_t k_tile_bytes = (int64_t)KVB * p.k_rs * 2;
      const int64_t k_left = ((int64_t)(Skv - 1) * p.k_rs + 128) * 2 - (int64_t)(t + 3) * k_tile_bytes;
      const int64_t v_left = (int64_t)128 * p.vt_rs * 2 - (int64_t)(t + 2) * KVB * 2;
      const uint64_t kt = uniform64((const char*)K + (int64_t)(t + 3) * k_tile_bytes);
      const uint64_t vtb = uniform64((const char*)VT + (int64_t)(t + 2) * KVB * 2);
      const int kd0 = sreg((int)(uint32_t)kt), kd1 = sreg((int)(uint32_t)(kt >> 32) & 0xffff), kd2 = sreg((int)(uint32_t)(k_left > 0 ? k_left : 0));
      const int vd0 = sreg((int)(uint32_t)vtb), vd1 = sreg((int)(uint32_t)(vtb >> 32) & 0xffff), vd2 = sreg((int)(uint32_t)(v_left > 0 ? v_left : 0));
      const int d3 = sreg(0x00020000);
      const int qvo0 = (int)(((int64_t)min(x.q_row, Sq - 1) * p.q_rs + x.h2 * 8) * 2);
      const int qvo1 = (int)(((int64_t)min(x.q_row + 32, Sq - 1) * p.q_rs + x.h2 * 8) * 2);
      const uint64_t qbs = uniform64(Q);
      const int kstep = sreg((int)k_tile_bytes), tend_s = sreg(tend);
      const int wk = sreg((int)kl + wave * 1024), wv = sreg((int)vl + wave * 1024);
      const float c_s = __builtin_bit_cast(float, sreg(__builtin_bit_cast(int, c)));
      const float negmc0 = -m_run[0] * c, negmc1 = -m_run[1] * c;
      int ts = sreg(t), code;
      asm volatile(ALG_ATTN128_Q64_LOOP_ASM
                   : ALG_ATTN128_Q64_O_OPERANDS(oa), [l0] "+v"(l_run[0]), [l1] "+v"(l_run[1]), [t] "+s"(ts), [code] "=&s"(code)
                   : [lk0] "v"(lk[0]), [lk1] "v"(lk[1]), [lk2] "v"(lk[2]), [lk3] "v"(lk[3]), [lk4] "v"(lk[4]), [lk5] "v"(lk[5]),
                     [lk6] "v"(lk[6]), [lk7] "v"(lk[7]), [lv0] "v"(lv[0]), [lv1] "v"(lv[1]), [lv2] "v"(lv[2]), [lv3] "v"(lv[3]),
                     [kvo0] "v"(kvo[0]), [kvo1] "v"(kvo[1]), [kvo2] "v"(kvo[2]), [kvo3] "v"(kvo[3]), [vvo0] "v"(vvo[0]),
                     [vvo1] "v"(vvo[1]), [vvo2] "v"(vvo[2]), [vvo3] "v"(vvo[3]),
                     [qvo0] "v"(qvo0), [qvo1] "v"(qvo1), [negmc0] "v"(negmc0), [negmc1] "v"(negmc1), [c] "s"(c_s),
                     [kd0] "s"(kd0), [kd1] "s"(kd1), [kd2] "s"(kd2), [kd3] "s"(d3), [vd0] "s"(vd0), [vd1] "s"(vd1), [vd2] "s"(vd2),
                     [vd3] "s"(d3), [qb] "s"(qbs), [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                   : "memory", "vcc", "scc", ALG_ATTN128_Q64_CLOBBERS);
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
    if (q_row < Sq) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 128;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * x.h2;
          uint2 v;
          v.x = pack_bf2(oa[4 * qh + dt][4 * g] * inv, oa[4 * qh + dt][4 * g + 1] * inv);
          v.y = pack_bf2(oa[4 * qh + dt][4 * g + 2] * inv, oa[4 * qh + dt][4 * g + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
  if (tap && x.l31 == 0 && x.h2 == 0) {
    uint64_t* cp = p.clk + (size_t)(blockIdx.x >> 6) * 4;   // one workgroup owns a slot (block / 64 < slots)
    cp[0] = tap_c0, cp[1] = tap_r0, cp[2] = __builtin_readcyclecounter(), cp[3] = wall_clock64();
  }
}

}  // namespace a128q

// Returns ALG_OK when launched, 1 when this call is not covered (the caller goes on to attention128_pipe.hip / attention128.hip).
int flash_attn_d128_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                        int64_t o_rs, float scale, hipStream_t stream) {
  using namespace a128q;
  // ALG_ATTN128_Q64: 1 (default) = calls over at least POLICY_TILES KV tiles; 2 = every call the kernel can take; 3 = as 2 with the
  // statement switched off (every tile through the frame's C++ body: tests); 0 = off.
  const int enabled = opt(OPT_ATTN128_Q64);
  const int n_tiles = (Skv + KVB - 1) / KVB;
  if (!enabled || n_tiles < (enabled == 1 ? POLICY_TILES : MIN_TILES)) return 1;
  // 31-bit BYTE offsets inside one (batch, head) for the DMA's lane offsets; V^T rows cover whole 64-key tiles
  if ((int64_t)(Skv + 64) * k_rs * 2 >= (1ll << 31) || (int64_t)129 * vt_rs * 2 >= (1ll << 31) ||
      (int64_t)Sq * q_rs * 2 >= (1ll << 31))
    return 1;
  if (vt_rs < (int64_t)((Skv + KVB - 1) / KVB) * KVB) return 1;
  static PerDeviceOnce attr_set;
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
    if (hipFuncSetAttribute((const void*)flash_attn_d128_q64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess)
      return 1;
    device_mark(attr_set, dev_slot);
  }
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + NW * QW - 1) / (NW * QW);
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.clk = clock_tap_for((hipStream_t)stream, &p.clk_slots);
  p.use_statement = enabled != 3;
  const int64_t grid = (int64_t)((batch * heads + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) return 1;
  hipLaunchKernelGGL(flash_attn_d128_q64_kernel, dim3((unsigned)grid), dim3(NW * 64), LDS_BYTES, stream, p);
  return check_launch("alg_flash_attn_d128");
}

}  // namespace alg
