// Flash attention forward, head_dim 64, the long self-attention form of the CogVideoX DiT on v_mfma_f32_16x16x32_bf16 (round 6).
//
// Why another d = 64 kernel: under the 1400 W package cap a register-only loop of the 16x16x32 bf16 MFMA sustains ~10 % more than
// the 32x32x16 loop on the same box (scripts/micro/mfma_shape.hip, profiles/r6_mfma_shape_power.txt: half the accumulator words per
// FLOP through the register file, the part clocks higher), and the attention of this build is power-bound.
// Same construction as attention.hip's flash_attn_d64_pipe_kernel<8> -- one 256-query unit per 8-wave workgroup, 32 queries per
// wave, K / V^T tiles of 64 through four-slot LDS-DMA rings, the steady state ONE generated asm statement with PV(t-1) / QK(t+1) /
// softmax(t) interleaved (attn64_m16_loop.inc, scripts/gen_attn_m16.py), pre-scaled scores with the running offset snapped to zero
// on tile 0, lazy running max, the same collective protocol inside and outside the statement -- with these differences:
//   * a wave's 32 queries are TWO 16-query blocks; S^T = K Q^T as 16 x 16 blocks: a lane (r = lane & 15, g = lane >> 4) holds, per
//     block (j, u) of a 64-key tile and query block qb, the four keys 32 j + 16 (g >> 1) + 8 u + 4 (g & 1) + e of query 16 qb + r.
//     The K fragment of block (j, u) therefore takes the tile's key rows {0..7, 16..23} + 32 j + 8 u -- chosen so that the two S
//     blocks (j, 0), (j, 1) of a lane, packed to bf16, ARE the B operand of the PV MFMA over keys [32 j, 32 j + 32) in the order
//     V^T is stored in (kv index bits 2 <-> 3 swapped per 16: what the V^T projection writes for every d = 64 kernel).  P never
//     leaves its registers; the global layouts of Q, K, V^T and O are those of alg_flash_attn_d64.
//   * two running row sums (and offsets) per lane; a row's total is the sum over its four lanes g = 0..3.
//   * the K tile's 16-byte chunk XOR in the LDS is swk(row) = ((row & 7) >> 1) | ((row >> 4) & 1) << 2: it depends on the row PART
//     of a fragment only, so a block is an immediate offset, and every 16-lane read group covers all sixteen bank groups.
// Main launch of the PRE-SCALED call (ALG_ATTN_Q_PRESCALED); the split-KV tail of alg_flash_attn_d64 stays on attention.hip's
// kernel.  Selected by ALG_ATTN_PP=7 (alg_hip.h).  Reference call site: the SDPA inside the CogVideoX blocks behind
// /root/reference/pipeline_cogvideox_image2video_lowpass.py:1082-1090.
#include <stdlib.h>

#include "common.h"
#include "attn64_m16_loop.inc"

namespace alg {
namespace a64m {

constexpr int NW = 8;
constexpr int KVB = 64;
constexpr int TILE = KVB * 64 * 2;       // 8 KiB: K tile = V^T tile
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

template <bool B>
struct BoolConst { static constexpr bool value = B; };

__device__ __forceinline__ int swk(int row) { return ((row & 7) >> 1) | (((row >> 4) & 1) << 2); }

struct Lane {
  int lane, r15, g4, tid, srow, kslot, vslot;
  int q_row[2];
  int lk[2], lv[2];     // byte offset of the lane's K / V^T fragment inside a tile, per k-step / key half
};

// S^T blocks of one 64-key tile: s[2 j + u][qb] = K(block j, u) Q(qb)^T
__device__ __forceinline__ void qk_tile(const char* Ks, const bf16x8 (&qf)[2][2], const Lane& c, f32x4 (&s)[4][2]) {
#pragma unroll
  for (int ju = 0; ju < 4; ++ju) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) s[ju][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(Ks + c.lk[ks] + (32 * (ju >> 1) + 8 * (ju & 1)) * 128);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s[ju][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s[ju][qb], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void mask_tail(f32x4 (&s)[4][2], int kv0, int S, int g4) {
#pragma unroll
  for (int ju = 0; ju < 4; ++ju)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kv = kv0 + 32 * (ju >> 1) + 16 * (g4 >> 1) + 8 * (ju & 1) + 4 * (g4 & 1) + e;
      if (kv >= S) s[ju][0][e] = -INFINITY, s[ju][1][e] = -INFINITY;
    }
}

// softmax_tile_zero of attention.hip for two queries per lane: pre-scaled scores, the offset snapped to zero when tile 0 allows it,
// lazy running max (the exact path runs for the WAVE when any row sum leaves [0, 2^80)).  pf[qb][j]: the PV operand over keys
// [32 j, 32 j + 32) = (block (j, 0) registers 0..3, block (j, 1) registers 0..3) packed to bf16.
__device__ __forceinline__ void softmax_tile_zero(const f32x4 (&s)[4][2], float (&m_run)[2], float (&l_run)[2], f32x4 (&oa)[4][2],
                                                  bf16x8 (&pf)[2][2]) {
  typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
  auto probs = [&](auto sub_c, int qb, float m) -> float {
    constexpr bool SUB = decltype(sub_c)::value;
    float psum = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // k = 2 u + (e >> 1)
        const float x0 = s[2 * j + (k >> 1)][qb][2 * (k & 1)], x1 = s[2 * j + (k >> 1)][qb][2 * (k & 1) + 1];
        const float p0 = __builtin_amdgcn_exp2f(SUB ? x0 - m : x0);
        const float p1 = __builtin_amdgcn_exp2f(SUB ? x1 - m : x1);
        pk.u[k] = pack_bf2(p0, p1);
        psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[k]), __builtin_bit_cast(bf2v, 0x3f803f80u), psum, false);
      }
      pf[qb][j] = pk.v;
    }
    return psum;
  };
  float psum[2];
  if (__all(m_run[0] == 0.0f && m_run[1] == 0.0f)) {
    psum[0] = probs(BoolConst<false>{}, 0, 0.0f);
    psum[1] = probs(BoolConst<false>{}, 1, 0.0f);
  } else {
    psum[0] = probs(BoolConst<true>{}, 0, m_run[0]);
    psum[1] = probs(BoolConst<true>{}, 1, m_run[1]);
  }
  if (__any(!(psum[0] < ALG_LAZY_SUM_LIMIT) || !(psum[1] < ALG_LAZY_SUM_LIMIT))) {   // 2^80; also inf (tile 0: m = -inf) and NaN
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mt = s[0][qb][0];
#pragma unroll
      for (int ju = 0; ju < 4; ++ju)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[ju][qb][e]);
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      float m_new = fmaxf(m_run[qb], mt);
      if (m_run[qb] == -INFINITY && fabsf(m_new) < 64.0f) m_new = 0.0f;  // tile 0: snap the offset to zero when it is safe
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      m_run[qb] = m_new;
      l_run[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 4; ++e) oa[db][qb][e] *= alpha;
      psum[qb] = probs(BoolConst<true>{}, qb, m_run[qb]);
    }
  }
  l_run[0] += psum[0];
  l_run[1] += psum[1];
}

// O^T += V^T P^T for one 64-key tile
__device__ __forceinline__ void pv_tile(const char* Vs, const bf16x8 (&pf)[2][2], const Lane& c, f32x4 (&oa)[4][2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const bf16x8 vf = *(const bf16x8*)(Vs + c.lv[j] + db * 2048);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) oa[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][j], oa[db][qb], 0, 0, 0);
    }
}

__global__ __launch_bounds__(NW * 64) void flash_attn_d64_m16_kernel(const P p) {
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];
  char* const k_ring = smem;
  char* const v_ring = smem + 4 * TILE;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nbh = p.batch * p.heads;
  int bh, qb0;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int slot = idx / p.q_blocks;
    qb0 = idx - slot * p.q_blocks;
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

  f32x4 oa[4][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) oa[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};

  // Everything lane-derived is rebuilt from a fresh lane id in front of and behind the statement (attention.hip: values live
  // across it compete with its 32 O operands for v[0:25])
  auto make_lane = [&](int lane) -> Lane {
    Lane c;
    c.lane = lane, c.r15 = lane & 15, c.g4 = lane >> 4, c.tid = wave * 64 + lane;
    c.srow = c.tid >> 3;
    c.kslot = (c.tid & 7) ^ swk(c.srow);
    c.vslot = (c.tid & 7) ^ ((c.tid >> 4) & 7);
    c.q_row[0] = qb0 * (NW * 32) + wave * 32 + c.r15, c.q_row[1] = c.q_row[0] + 16;
    const int rowpart = c.r15 + 8 * (c.r15 >> 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      c.lk[ks] = rowpart * 128 + (((4 * ks + c.g4) ^ swk(rowpart)) * 16);
      c.lv[ks] = c.r15 * 128 + (((4 * ks + c.g4) ^ ((c.r15 >> 1) & 7)) * 16);
    }
    return c;
  };
  auto fresh_lane = [&]() -> int {
    int z = 0;
    asm volatile("" : "+s"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  };
  auto stage_k = [&](const Lane& c, int t) {
    const bf16_t* ks = K + (int64_t)min(t * KVB + c.srow, S - 1) * p.q_rs + c.kslot * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + (t & 3) * TILE + wave * 1024), 16, 0, 0);
  };
  auto stage_v = [&](const Lane& c, int t) {
    __builtin_amdgcn_global_load_lds((gptr_t)(VT + (int64_t)c.srow * p.vt_rs + c.vslot * 8 + min(t, T - 1) * KVB),
                                     (lptr_t)(v_ring + (t & 3) * TILE + wave * 1024), 16, 0, 0);
  };
  // iterations [t, t_end) in the straight form: protocol (unless the first one's is already done), QK(t) -> softmax -> PV(t)
  auto straight = [&](const Lane& c, int t, int t_end, bool top_done) {
    bf16x8 qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const bf16_t* qp = Q + (int64_t)min(c.q_row[qb], S - 1) * p.q_rs + c.g4 * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + ks * 32);
    }
    for (; t < t_end; ++t) {
      if (!top_done) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // all but the previous iteration's two DMA pieces
        __syncthreads();
        stage_k(c, t + 3);   // past the end the source rows are clamped (K) / lie in the padded pitch (V^T): the DMA count per
        stage_v(c, t + 2);   // iteration must not depend on t, the counted wait above relies on it
      }
      top_done = false;
      f32x4 s[4][2];
      qk_tile(k_ring + (t & 3) * TILE, qf, c, s);
      if (ragged && t == T - 1) mask_tail(s, t * KVB, S, c.g4);
      bf16x8 pf[2][2];
      softmax_tile_zero(s, m_run, l_run, oa, pf);
      pv_tile(v_ring + (t & 3) * TILE, pf, c, oa);
    }
  };

  // the statement only runs iterations t whose DMA target K(t + 3) is a whole tile and whose tile t + 1 needs no mask
  const int tend = ragged ? T - 4 : T - 3;
  int t = 1;
  bool top_done = false;
  {
    const Lane c = make_lane(fresh_lane());
    stage_k(c, 0);
    stage_k(c, 1);
    stage_v(c, 0);
    stage_v(c, 0);       // (filler: two DMAs per batch)
    stage_k(c, 2);       // the batch "iteration -1" would have issued: K(2), V(1)
    stage_v(c, 1);
    straight(c, 0, 1, false);     // tile 0: establishes the running offsets (snapped to zero when its scores allow)
  }
  if (1 + 4 <= tend && __all(m_run[0] == 0.0f && m_run[1] == 0.0f)) {
    const Lane c = make_lane(fresh_lane());
    auto sreg = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
    auto uniform64 = [](const void* ptr) -> uint64_t {
      const uint64_t v = (uint64_t)(uintptr_t)ptr;
      return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint32_t kl = (uint32_t)(uintptr_t)(lptr_t)k_ring, vl = (uint32_t)(uintptr_t)(lptr_t)v_ring;
    const int lk0 = kl + c.lk[0], lk1 = kl + c.lk[1], lv0 = vl + c.lv[0], lv1 = vl + c.lv[1];
    int kvo0 = (int)(((int64_t)((t + 3) * KVB + c.srow) * p.q_rs + c.kslot * 8) * 2);
    int vvo0 = (int)(((int64_t)c.srow * p.vt_rs + c.vslot * 8 + (t + 2) * KVB) * 2);
    const int qvo0 = (int)(((int64_t)min(c.q_row[0], S - 1) * p.q_rs + c.g4 * 8) * 2);
    const int qvo1 = (int)(((int64_t)min(c.q_row[1], S - 1) * p.q_rs + c.g4 * 8) * 2);
    const uint64_t kb = uniform64(K), vb = uniform64(VT), qbs = uniform64(Q);
    const int kstep = sreg((int)(KVB * p.q_rs * 2)), tend_s = sreg(tend);
    const int wk = sreg((int)kl + wave * 1024), wv = sreg((int)vl + wave * 1024);
    int ts = sreg(t), code;
    float o[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i] = oa[i >> 3][(i >> 2) & 1][i & 3];
    float l0 = l_run[0], l1 = l_run[1];
    asm volatile(ALG_ATTN_M16_LOOP_ASM
                 : ALG_ATTN_M16_O_OPERANDS(o), [l0] "+v"(l0), [l1] "+v"(l1), [t] "+s"(ts), [code] "=&s"(code), [kvo0] "+v"(kvo0),
                   [vvo0] "+v"(vvo0)
                 : [lk0] "v"(lk0), [lk1] "v"(lk1), [lv0] "v"(lv0), [lv1] "v"(lv1), [qvo0] "v"(qvo0), [qvo1] "v"(qvo1), [kb] "s"(kb),
                   [vb] "s"(vb), [qb] "s"(qbs), [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                 : "memory", "vcc", "scc", ALG_ATTN_M16_CLOBBERS);
#pragma unroll
    for (int i = 0; i < 32; ++i) oa[i >> 3][(i >> 2) & 1][i & 3] = o[i];
    l_run[0] = l0, l_run[1] = l1;
    t = ts;
    top_done = code != 0;   // 1: iteration t's protocol is done, softmax(t) is not: tile t is redone below
  }
  const Lane c = make_lane(fresh_lane());
  straight(c, t, T, top_done);   // the tiles behind the statement (or all of them but tile 0)

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    if (c.q_row[qb] < S) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)c.q_row[qb] * p.o_rs + h * 64;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 v;
        v.x = pack_bf2(oa[db][qb][0] * inv, oa[db][qb][1] * inv);
        v.y = pack_bf2(oa[db][qb][2] * inv, oa[db][qb][3] * inv);
        *(uint2*)(op + 16 * db + 4 * c.g4) = v;
      }
    }
  }
  if (tap && c.lane == 0) {
    uint64_t* cp = p.clk + (size_t)(blockIdx.x >> 6) * 4;   // one workgroup owns a slot (block / 64 < slots)
    cp[0] = tap_c0, cp[1] = tap_r0, cp[2] = __builtin_readcyclecounter(), cp[3] = wall_clock64();
  }
}

}  // namespace a64m

// Main launch of the pre-scaled call on `blocks` workgroups.  Returns ALG_OK when launched, 1 when this call is not covered (the
// caller launches attention.hip's main kernel instead), < 0 on error.
int flash_attn_d64_m16(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S, int q_blocks,
                       int64_t q_bs, int64_t q_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs, int64_t o_rs, unsigned blocks,
                       hipStream_t stream) {
  using namespace a64m;
  if (opt(OPT_ATTN_PP) != 7 || (S + KVB - 1) / KVB < MIN_TILES || blocks == 0) return 1;
  if (q_blocks != (S + NW * 32 - 1) / (NW * 32)) return 1;
  if ((int64_t)(S + 4 * KVB) * q_rs * 2 >= (1ll << 31) || (int64_t)65 * vt_rs * 2 >= (1ll << 31)) return 1;   // 31-bit byte offsets
  if (vt_rs < (int64_t)((S + KVB - 1) / KVB) * KVB) return 1;
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.S = S; p.q_blocks = q_blocks;
  p.q_bs = q_bs; p.q_rs = q_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.clk = clock_tap_for((hipStream_t)stream, &p.clk_slots);
  hipLaunchKernelGGL(flash_attn_d64_m16_kernel, dim3(blocks), dim3(NW * 64), 0, stream, p);
  return check_launch("alg_flash_attn_d64");
}

}  // namespace alg
