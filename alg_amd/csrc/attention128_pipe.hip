// Flash attention forward, head_dim 128, the long self-attention form of the Wan / HunyuanVideo DiTs: the pipelined kernel
// (round 3).  The d = 128 sibling of flash_attn_d64_pipe_kernel (attention.hip; read that header and scripts/gen_attn_pipe.py
// for the reasoning): the steady-state KV loop is ONE generated asm statement (attn128_pipe_loop.inc,
// scripts/gen_attn128_pipe.py) in which every MFMA is followed, in program order of the same wave, by a slice of the softmax
// of the previous tile and one fragment read -- PV(t-1), QK(t+1) and softmax(t) software-pipelined.  This file is the frame:
// attention128.hip's workgroup -> (head, q block) order, operand layouts, swizzles and lazy running max, a C++ loop for tile 0
// (where the running max is established), the last tiles (ragged tail) and any tile the statement refuses (row sum outside
// [0, 2^80): exact max / rescale), under the statement's collective protocol
//     top of iteration t:  s_waitcnt vmcnt(8); s_barrier; DMA K(t+3) -> K slot (t+3) & 3, V^T(t+2) -> V slot (t+2) & 3
// so the waves of a workgroup may be inside or outside the statement independently.  Four waves x 32 queries per workgroup,
// one wave per SIMD (the statement names v[64:169] and a[0:127]; O^T travels in 64 operands); K / V^T through two four-slot
// rings of 16 KiB tiles (128 KiB of LDS).  Used for non-causal, ungrouped attention over at least MIN_TILES KV tiles.  DEFAULT
// (ALG_ATTN128_PIPE=0 switches back to attention128.hip): C3 +4.4 %, C4 +5.9 %, C5 +4.8 % frames/s
// (profiles/r3_attention128_pipe_ab.txt); 0 of 600 fp8 C5 forwards in three fresh processes differ run to run
// (profiles/r3_attention128_pipe_determinism.jsonl) -- the test the 64-query kernel fails.
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "attn128_pipe_loop.inc"

namespace alg {
namespace a128p {

constexpr int NW = 4;
constexpr int KVB = 64;
constexpr int TILE = 16384;               // K tile = V^T tile
constexpr int LDS_BYTES = 8 * TILE;       // 4 K slots + 4 V^T slots
constexpr int MIN_TILES = 12;

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
};

__global__ __launch_bounds__(NW * 64) void flash_attn_d128_pipe_kernel(const P p) {
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
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 128;
  const bf16_t* K = p.k + (int64_t)b * p.k_bs + h * 128;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 128 * p.vt_rs;
  const int T = (Skv + KVB - 1) / KVB;
  const bool ragged = (Skv & (KVB - 1)) != 0;
  const float c = p.scale_log2;
  f32x16 oa[4];
#pragma unroll
  for (int i = 0; i < 64; ++i) oa[i >> 4][i & 15] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  // everything lane-derived is rebuilt from a freshly laundered lane id by each of the three phases (in front of, inside the
  // operand set-up of, and behind the statement), so that none of it is live across the statement next to its 64 O operands
  struct LaneCtx {
    int l31, h2, tid, k_row, k_slot, v_row, v_slot, q_row, k_row_off, k_sw, v_row_off, v_sw;
  };
  auto make_ctx = [&](int lane) -> LaneCtx {
    LaneCtx x;
    x.l31 = lane & 31, x.h2 = lane >> 5, x.tid = wave * 64 + lane;
    x.k_row = x.tid >> 4, x.k_slot = (x.tid & 15) ^ ((x.tid >> 4) & 15);      // + 16 rows per DMA piece
    x.v_row = x.tid >> 3, x.v_slot = (x.tid & 7) ^ ((x.tid >> 4) & 7);        // + 32 d-rows per DMA piece
    x.q_row = qb * (NW * 32) + wave * 32 + x.l31;
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
  // iterations [t, t_end) in the straight form (attention128.hip's tile body on the four-slot rings)
  auto straight = [&](const LaneCtx& x, int t, int t_end, bool top_done) {
    bf16x8 qf[8];
    const bf16_t* qp = Q + (int64_t)min(x.q_row, Sq - 1) * p.q_rs + x.h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    for (; t < t_end; ++t) {
      if (!top_done) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but the previous iteration's eight DMAs
        __syncthreads();
        stage_k(x, t + 3);   // (past the end: clamped sources; the DMA count per iteration must not depend on t)
        stage_v(x, t + 2);
      }
      top_done = false;
      const char* Ks = k_ring + (t & 3) * TILE + x.k_row_off;
      const char* Vs = v_ring + (t & 3) * TILE + x.v_row_off;
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
      typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
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
              psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]),
                                                    __builtin_bit_cast(bf2v, 0x3f803f80u), psum, false);
            }
            pf[sub * 2 + g] = pk.v;
          }
        return psum;
      };
      float psum = probs(m_run * c);
      if (__any(!(psum < ALG_LAZY_SUM_LIMIT))) {  // 2^80; also inf / NaN
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
        {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
          mt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oa[dt] *= alpha;
        psum = probs(m_run * c);
      }
      l_run += psum;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 vf = *(const bf16x8*)(Vs + dt * 4096 + (((2 * kk + x.h2) ^ x.v_sw) * 16));
          oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], oa[dt], 0, 0, 0);
        }
    }
  };

  // the statement only runs iterations t whose DMA target K(t + 3) is a whole tile and whose tile t + 1 needs no mask
  const int tend = ragged ? T - 4 : T - 3;
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
    straight(x, 0, 1, false);     // tile 0: establishes the running max
  }
  if (1 + 4 <= tend && __all(m_run > -3.0e38f)) {
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
    int kvo[4], vvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kvo[i] = (int)(((int64_t)((t + 3) * KVB + x.k_row + 16 * i) * p.k_rs + x.k_slot * 8) * 2);
      vvo[i] = (int)(((int64_t)(x.v_row + 32 * i) * p.vt_rs + x.v_slot * 8 + (t + 2) * KVB) * 2);
    }
    const int qvo = (int)(((int64_t)min(x.q_row, Sq - 1) * p.q_rs + x.h2 * 8) * 2);
    const uint64_t kb = uniform64(K), vb = uniform64(VT), qbs = uniform64(Q);
    const int kstep = sreg((int)(KVB * p.k_rs * 2)), tend_s = sreg(tend);
    const int wk = sreg((int)kl + wave * 1024), wv = sreg((int)vl + wave * 1024);
    const float c_s = __builtin_bit_cast(float, sreg(__builtin_bit_cast(int, c)));
    const float negmc = -m_run * c;
    int ts = sreg(t), code;
    float o[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] = oa[i >> 4][i & 15];
    asm volatile(ALG_ATTN128_PIPE_LOOP_ASM
                 : ALG_ATTN128_PIPE_O_OPERANDS(o), [l] "+v"(l_run), [t] "+s"(ts), [code] "=&s"(code), [kvo0] "+v"(kvo[0]),
                   [kvo1] "+v"(kvo[1]), [kvo2] "+v"(kvo[2]), [kvo3] "+v"(kvo[3]), [vvo0] "+v"(vvo[0]), [vvo1] "+v"(vvo[1]),
                   [vvo2] "+v"(vvo[2]), [vvo3] "+v"(vvo[3])
                 : [lk0] "v"(lk[0]), [lk1] "v"(lk[1]), [lk2] "v"(lk[2]), [lk3] "v"(lk[3]), [lk4] "v"(lk[4]), [lk5] "v"(lk[5]),
                   [lk6] "v"(lk[6]), [lk7] "v"(lk[7]), [lv0] "v"(lv[0]), [lv1] "v"(lv[1]), [lv2] "v"(lv[2]), [lv3] "v"(lv[3]),
                   [qvo] "v"(qvo), [negmc] "v"(negmc), [c] "s"(c_s), [kb] "s"(kb), [vb] "s"(vb), [qb] "s"(qbs),
                   [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                 : "memory", "vcc", "scc", ALG_ATTN128_PIPE_CLOBBERS);
#pragma unroll
    for (int i = 0; i < 64; ++i) oa[i >> 4][i & 15] = o[i];
    t = ts;
    top_done = code != 0;   // 1: iteration t's protocol is done, softmax(t) is not: tile t is redone below
  }
  const LaneCtx x = make_ctx(fresh_lane());
  straight(x, t, T, top_done);   // the tiles behind the statement (or all of them but tile 0)

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (x.q_row < Sq) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)x.q_row * p.o_rs + h * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * x.h2;
        uint2 v;
        v.x = pack_bf2(oa[dt][4 * g] * inv, oa[dt][4 * g + 1] * inv);
        v.y = pack_bf2(oa[dt][4 * g + 2] * inv, oa[dt][4 * g + 3] * inv);
        *(uint2*)(op + d) = v;
      }
  }
}

}  // namespace a128p

// 0 = launched, 1 = not covered (the caller falls back to attention128.hip), < 0 = error
int flash_attn_d128_pipe(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                         int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                         int64_t o_rs, float scale, hipStream_t stream) {
  using namespace a128p;
  if (!opt(OPT_ATTN128_PIPE) || (Skv + KVB - 1) / KVB < MIN_TILES) return 1;   // default since round 3; ALG_ATTN128_PIPE=0: attention128.hip
  // 31-bit BYTE offsets inside one (batch, head) for the DMA's lane offsets; V^T rows cover whole 64-key tiles
  if ((int64_t)(Skv + 64) * k_rs * 2 >= (1ll << 31) || (int64_t)129 * vt_rs * 2 >= (1ll << 31) ||
      (int64_t)Sq * q_rs * 2 >= (1ll << 31))
    return 1;
  if (vt_rs < (int64_t)((Skv + KVB - 1) / KVB) * KVB) return 1;
  static PerDeviceOnce attr_set;
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
    if (hipFuncSetAttribute((const void*)flash_attn_d128_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess)
      return 1;
    device_mark(attr_set, dev_slot);
  }
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + NW * 32 - 1) / (NW * 32);
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.scale_log2 = scale * 1.4426950408889634f;
  const int64_t grid = (int64_t)((batch * heads + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) return 1;
  hipLaunchKernelGGL(flash_attn_d128_pipe_kernel, dim3((unsigned)grid), dim3(NW * 64), LDS_BYTES, stream, p);
  return check_launch("alg_flash_attn_d128");
}

}  // namespace alg
