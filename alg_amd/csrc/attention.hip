// Flash attention forward (full softmax, no mask), head_dim 64, bf16 in/out, fp32 accumulate, for the
// joint [text; video] sequence of the video DiT (17,776 tokens at the north-star config).
//
// Wave-level formulation (64-wide waves, v_mfma_f32_32x32x16_bf16):
//   S^T = K Q^T   : A = K tile rows (kv), B = Q^T.  In the 32x32 C layout a lane then owns ONE query column
//                   (q = lane & 31) and 16 kv rows per sub-tile, so the softmax running max / sum / rescale
//                   are per-lane scalars; the two half-waves own complementary kv rows and exchange only the
//                   tile max (one cross-lane op per KV tile).
//   O^T = V^T P^T : A = V^T (rows = d), B = P^T taken straight from the S^T accumulators (exp -> bf16 pack),
//                   no LDS round trip for P.  The contraction index order of B is whatever the C layout hands
//                   out (kv = 16g + 8(j>>2) + 4h + (j&3)); V^T is stored by its producer GEMM with index bits
//                   2 and 3 swapped so one ds_read_b128 yields the matching A fragment.
// Workgroup = 8 waves x 32 queries = 256 queries of one (batch, head); KV tiles of 64 stream through an LDS
// ring with 16-byte global_load_lds; the bank swizzle sits on the source address (same scheme as gemm.hip).
// Workgroups are ordered so that one XCD works on one head at a time (K/V of a head = 4.5 MB, re-read by the
// 70 query blocks of that head out of the XCD's L2).
//
// Variants (template parameter, selected per call; ALG_ATTN_VARIANT env overrides for A/B runs):
//   0  straight loop: QK^T -> softmax -> PV per tile
//   1  + the O / l rescale is skipped (exactly: alpha == 1) when no lane's running max grew
//   2  + software pipelined: QK^T of tile t+1 is issued before the softmax of tile t, so the MFMA pipe works
//        under the VALU-heavy softmax; K runs one tile ahead of V through a 3-slot ring
//   3/4  "lean" softmax (pre-scaled Q, -m folded into the accumulator chain, packed row sums) -- see below
//   5  variant 1 with 4-wave workgroups;  6/7  64 queries per wave (8 / 4 waves per workgroup)
//   8/9  two KV tiles per barrier (9: + s_setprio around the MFMA clusters);  12  16-wave workgroups
//   13  variant 1 with the ragged tail tile peeled (no per-tile v_cndmask) and packed fma / row sums (-30 % VALU)
//   14  13 + explicitly staged fragments (8 K reads in one batch, V^T reads issued before / under the softmax) and a
//       v_permlane32_swap max exchange: no exposed LDS round trip left in the loop
//   15/16  "duo": two 32-query streams per wave sharing every fragment read, one stream's MFMAs interleaved with the
//       other's softmax by sched_group_barrier (8 / 4 waves per workgroup)
//   32  variant 1 with the row sums taken by v_dot2c_f32_bf16 from the packed P pairs (16 instructions per tile instead
//       of 32 adds; sums the bf16-rounded probabilities the PV MFMA uses): 925 -> 945 TFLOP/s on the same box
//   33  (DEFAULT) 32 + lazy running max (softmax_tile_lazy): no tile max in the common path, the exact max / rescale
//       path runs only when a row sum leaves [0, 2^80): 919 -> 990 TFLOP/s on the same box; the split-KV tail launch is
//       built on it
//   39/40  the duo kernels (15/16) with the lazy softmax; the exact-max fix-up sits between the scheduling regions:
//       828 -> 995 / 1044 TFLOP/s (8 / 4 waves), i.e. on par with 33 (1027 on the same box); the interleave granularity
//       (6..16 VALU per MFMA) moves it by < 2 %
//   41  the kernel behind alg_flash_attn_d64_ex(ALG_ATTN_Q_PRESCALED): variant 34's softmax on a Q that was scaled where it
//       was produced (alg_qk_norm_rope_scaled: no extra rounding); not selectable by ALG_ATTN_VARIANT
//   34  33 with Q pre-scaled by scale*log2(e) in registers and the offset snapped to zero when the first tile's max allows
//       it: p = exp2(s) with no per-score fma (+2 %; one more bf16 rounding of q, so opt-in)
//   42/43  (round 3, ALG_ATTN_PP=1/2) variant 41's arithmetic with the two waves of a SIMD half a tile apart (ping-pong over
//       workgroup barriers: one wave's softmax under its partner's PV + QK MFMAs; 43 adds s_setprio around the MFMA phase).
//       Bit-identical to 41; measured 1097 / 1096 against 1103-1111 TFLOP/s, same clock (1.86 GHz) and power (1310-1320 W):
//       the phase order across waves is not what limits the kernel (profiles/r3_attention_d64_pingpong.txt)
// Measured on MI355X at the C2 shape (2 x 48 heads x 17,776 tokens, profiles/): 1: 860-915 TFLOP/s,
// 0: 875, 2: 830, 3: 867, 4: 861, 5: 836, 6: 804, 7: 599, 8: 899, 9: 893, 12: 860, 13: 880, 14: 865, 15: 835, 16: 875.
// All of them sit at 1225-1330 W with the clock pulled down to 1.9-2.2 GHz (profiles/r1_power_and_issue_rates.txt):
// the kernel is bound by energy per FLOP under the package power cap, so variants that only remove stalls or issue
// slots gain clock, not time.  The variants stay selectable (ALG_ATTN_VARIANT) for A/B runs and parity tests.
#include <stdlib.h>

#include "common.h"
#include "attn_pipe_loop.inc"

namespace alg {

constexpr int ATT_THREADS = 512;
constexpr int QB = 256;   // queries per workgroup
constexpr int KVB = 64;   // kv rows per tile
constexpr int ATT_TILE = KVB * 64 * 2;     // 8 KiB (K tile; V^T tile is the same size)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnP {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int batch, heads, S, q_blocks;
  int64_t q_bs, q_rs, vt_bs, vt_rs, o_bs, o_rs;
  float scale_log2;  // scale * log2(e)
  // split-KV tail (see alg_flash_attn_d64): the last `tail_units` (head, q block) units of every XCD are cut into
  // `tail_split` KV chunks of `tail_tiles` KV tiles each; chunk results go to a workspace and are merged by a second kernel
  int unit0, tail_units, tail_split, tail_tiles;
  float* ws_o;   // [8 * tail_units][tail_split][q rows per block][64] fp32, unnormalised
  float* ws_ml;  // [8 * tail_units][tail_split][q rows per block][2]: running max (raw score units), row sum
  int prio;      // 1: the younger half of an 8-wave workgroup (waves 4-7) runs at s_setprio 1 (ALG_ATTN_PRIO, A/B knob)
  uint64_t* clk;   // clock tap (calibrate.hip: alg_attn_clock_tap) or NULL: {cycles, wall} at start / end of every 64th workgroup
  int clk_slots;
};

struct Frag {
  int row_off;  // l31 * 128
  int sw;       // (l31 >> 1) & 7
  int h2;
};

// S^T sub-tiles of one 64-row K tile: s[sub] = K[sub] Q^T   (NOLDS: measurement only, A operand = the Q fragment)
template <bool NOLDS = false>
__device__ __forceinline__ void qk_tile(const char* Ks, const bf16x8 (&qf)[4], const Frag f, f32x16 (&s)[2]) {
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf =
          NOLDS ? qf[ks ^ 1] : *(const bf16x8*)(Ks + f.row_off + sub * 4096 + (((2 * ks + f.h2) ^ f.sw) * 16));
      s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[sub], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void mask_tail(f32x16 (&s)[2], int kv0, int S, int h2) {
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kv = kv0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
      if (kv >= S) s[sub][e] = -INFINITY;
    }
}

// online softmax of one tile's scores -> bf16 P fragments; updates m, l and rescales O when needed
// NOEXP (measurement only, wrong results): 1 = half of the exp2 replaced by the bare fma, 2 = all of them
typedef float f32x2p __attribute__((ext_vector_type(2)));
template <bool B>
struct BoolC { static constexpr bool value = B; };

// PK: the per-score fma and the row sums are written on float2 so they lower to v_pk_fma_f32 / v_pk_add_f32
// DOT2: the row sum comes from v_dot2c_f32_bf16 on the packed P pairs (one instruction per two scores instead of two adds);
//       it then sums the bf16-rounded probabilities -- the values the PV MFMA actually uses
template <bool SKIP, int NOEXP = 0, bool PK = false, bool DOT2 = false>
__device__ __forceinline__ void softmax_tile(const f32x16 (&s)[2], float c, float& m_run, float& l_run,
                                             f32x16 (&o_acc)[2], bf16x8 (&pf)[4]) {
  float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
  for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  const bool grow = mt > m_run;
  if (!SKIP || __any(grow)) {
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
  }
  const float mc = m_run * c;
  if (PK) {
    const f32x2p c2 = {c, c}, mc2 = {mc, mc};
    f32x2p ps2 = {0.0f, 0.0f};
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2p sv = {s[sub][8 * g + 2 * j], s[sub][8 * g + 2 * j + 1]};
          const f32x2p x = __builtin_elementwise_fma(sv, c2, -mc2);
          const f32x2p pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
          ps2 += pv;
          pk.u[j] = pack_bf2(pv.x, pv.y);
        }
        pf[sub * 2 + g] = pk.v;
      }
    l_run += ps2.x + ps2.y;
    return;
  }
  float psum = 0.0f;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = s[sub][8 * g + 2 * j] * c - mc, x1 = s[sub][8 * g + 2 * j + 1] * c - mc;
        const float p0 = NOEXP >= 2 ? x0 : __builtin_amdgcn_exp2f(x0);
        const float p1 = NOEXP >= 1 ? x1 : __builtin_amdgcn_exp2f(x1);
        pk.u[j] = pack_bf2(p0, p1);
        if (DOT2) {
          typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
          psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]), __builtin_bit_cast(bf2v, 0x3f803f80u),
                                                 psum, false);
        } else {
          psum += p0 + p1;  // the row sum uses the unrounded fp32 probabilities (as the math SDPA path does)
        }
      }
      pf[sub * 2 + g] = pk.v;
    }
  l_run += psum;
}

// Lazy running max (variant 33).  Softmax is invariant to the subtracted offset, and fp32 / bf16 keep their relative
// precision at any magnitude, so the offset only has to keep exp2 inside the exponent range: the probabilities are formed
// against the CURRENT m (no tile max: 23 VALU instructions saved per tile) and the exact path -- tile max, grow m, rescale
// O and l, recompute -- runs only when a tile's row sum leaves [0, 2^80) (inf on the first tile, where m = -inf; NaN if
// a masked score meets m = -inf).  m is then always a max actually seen, so nothing that matters can underflow.
template <bool DOT2 = true>
__device__ __forceinline__ void softmax_tile_lazy(const f32x16 (&s)[2], float c, float& m_run, float& l_run,
                                                  f32x16 (&o_acc)[2], bf16x8 (&pf)[4]) {
  typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
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
          if (DOT2)
            psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]), __builtin_bit_cast(bf2v, 0x3f803f80u),
                                                   psum, false);
          else
            psum += p0 + p1;
        }
        pf[sub * 2 + g] = pk.v;
      }
    return psum;
  };
  float psum = probs(m_run * c);
  if (__any(!(psum < ALG_LAZY_SUM_LIMIT))) {  // 2^80; also true for inf and NaN
    float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
    psum = probs(m_run * c);
  }
  l_run += psum;
}

// Variant 34: Q is pre-scaled, so the scores arrive in log2 units and the only per-score VALU work left in the common path
// is exp2, the bf16 pack and the dot2 row sum: the offset is SNAPPED TO ZERO whenever the first tile's max lies in
// (-64, 64) (probabilities then span 2^-64 .. 2^64 at most before the lazy rescale threshold trips: harmless for fp32 /
// bf16), and p = exp2(s) needs no subtraction at all.  Rows whose scores are further out keep a non-zero offset and take
// the subtracting path; the exact max / rescale path is the lazy one of variant 33.
__device__ __forceinline__ void softmax_tile_zero(const f32x16 (&s)[2], float& m_run, float& l_run, f32x16 (&o_acc)[2],
                                                  bf16x8 (&pf)[4]) {
  typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
  auto probs = [&](auto sub_c, float m) -> float {
    constexpr bool SUB = decltype(sub_c)::value;
    float psum = 0.0f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = s[sub][8 * g + 2 * j], x1 = s[sub][8 * g + 2 * j + 1];
          const float p0 = __builtin_amdgcn_exp2f(SUB ? x0 - m : x0);
          const float p1 = __builtin_amdgcn_exp2f(SUB ? x1 - m : x1);
          pk.u[j] = pack_bf2(p0, p1);
          psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]), __builtin_bit_cast(bf2v, 0x3f803f80u),
                                                 psum, false);
        }
        pf[sub * 2 + g] = pk.v;
      }
    return psum;
  };
  float psum;
  if (__all(m_run == 0.0f))
    psum = probs(BoolC<false>{}, 0.0f);
  else
    psum = probs(BoolC<true>{}, m_run);
  if (__any(!(psum < ALG_LAZY_SUM_LIMIT))) {  // 2^80; also inf (first tile: m = -inf) and NaN
    float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float m_new = fmaxf(m_run, mt);
    if (m_run == -INFINITY && fabsf(m_new) < 64.0f) m_new = 0.0f;  // first tile: snap the offset to zero when it is safe
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
    psum = probs(BoolC<true>{}, m_run);
  }
  l_run += psum;
}

// O^T += V^T P^T for one 64-row tile
template <bool NOLDS = false>
__device__ __forceinline__ void pv_tile(const char* Vs, const bf16x8 (&pf)[4], const Frag f, f32x16 (&o_acc)[2]) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {  // kk = sub*2 + g : kv block [16 kk, 16 kk + 16)
      const bf16x8 vf =
          NOLDS ? pf[kk ^ 1] : *(const bf16x8*)(Vs + f.row_off + dt * 4096 + (((2 * kk + f.h2) ^ f.sw) * 16));
      o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], o_acc[dt], 0, 0, 0);
    }
}

// NW = waves per workgroup (8 or 4).  With 4 waves a workgroup puts ONE wave on each SIMD, so the waves that share a
// SIMD belong to different workgroups and are not phase-locked by the per-tile barrier (one runs MFMAs while the
// other runs its softmax); the price is that K/V^T tiles are staged once per 128 instead of 256 queries.
template <int VARIANT, int NW = 8, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64, (VARIANT == 2 ? 2 : 4)) void flash_attn_d64_kernel(const AttnP p) {
  static_assert(!SPLIT || ((VARIANT == 33 || VARIANT == 41) && NW == 8), "the split-KV tail is built on the default variants");
  constexpr int ROUNDS = NW >= 8 ? 1 : 8 / NW;  // DMA rounds per 8 KiB tile (one round = min(NW, 8) KiB)
  constexpr int DW = NW >= 8 ? 8 : NW;           // waves that issue DMA (a 16-wave workgroup only needs half)
  constexpr int K_SLOTS = VARIANT == 2 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char smem[(K_SLOTS + 2) * ATT_TILE];
  char* const k_ring = smem;
  char* const v_ring = smem + K_SLOTS * ATT_TILE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h2 = lane >> 5;

  // ---- workgroup -> (batch*head, q block): XCD x walks heads x, x+8, ... ----
  const int nbh = p.batch * p.heads;
  int bh, qb;
  int part = 0, chunk = 0;  // SPLIT: workspace slot of this (unit, chunk)
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int idx = bid >> 3;
    if (SPLIT) {  // the tail launch: block j of an XCD = chunk j % split of tail unit j / split
      chunk = idx % p.tail_split;
      const int u = idx / p.tail_split;
      part = xcd * p.tail_units + u;
      idx = p.unit0 + u;
    }
    const int slot = idx / p.q_blocks;
    qb = idx - slot * p.q_blocks;
    bh = slot * 8 + xcd;
    if (bh >= nbh) return;
  }
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int S = p.S;
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* K = p.k + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 64 * p.vt_rs;

  // ---- Q^T fragments (B operand): lane (q = l31, h2) holds Q[q][16 ks + 8 h2 .. +8] ----
  const int q_row = qb * (NW * 32) + wave * 32 + l31;
  bf16x8 qf[4];
  {
    const bf16_t* qp = Q + (int64_t)min(q_row, S - 1) * p.q_rs + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    if (VARIANT == 34) {  // scores come out of the MFMA in log2 units: q * (scale * log2 e), one more bf16 rounding of q
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        union { bf16x8 v; uint32_t u[4]; } raw, sc;
        raw.v = qf[ks];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          sc.u[j] = pack_bf2(__uint_as_float(raw.u[j] << 16) * p.scale_log2,
                             __uint_as_float(raw.u[j] & 0xffff0000u) * p.scale_log2);
        qf[ks] = sc.v;
      }
    }
  }

  // ---- DMA sources: ROUNDS 16-B pieces of K and of V^T per thread per tile ----
  // per-lane base pointers for tile 0; a tile adds a wave-uniform byte offset (scalar multiply), and the row clamp
  // (rows >= S re-read row S-1, masked later) only exists on the last, ragged tile
  const int srow = tid >> 3;                        // K: kv row, V^T: d row   (+ NW*8 per round)
  const int sslot = (tid & 7) ^ ((tid >> 4) & 7);   // (row >> 1) & 7 does not depend on the round
  const bf16_t* k_src[ROUNDS];
  const bf16_t* v_src[ROUNDS];
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    k_src[i] = K + (int64_t)min(srow + i * DW * 8, S - 1) * p.q_rs + sslot * 8;
    v_src[i] = VT + (int64_t)(srow + i * DW * 8) * p.vt_rs + sslot * 8;
  }
  const int64_t k_tile_stride = (int64_t)KVB * p.q_rs;
  const int last_tile = (S + KVB - 1) / KVB - 1;
  const bool ragged_src = (S & (KVB - 1)) != 0;
  auto stage_k = [&](int slot, int kv0) {
    if (NW > 8 && wave >= 8) return;
    const int t = kv0 / KVB;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      const bf16_t* ks = k_src[i] + t * k_tile_stride;
      if (ragged_src && t == last_tile) ks = K + (int64_t)min(kv0 + srow + i * DW * 8, S - 1) * p.q_rs + sslot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + slot * ATT_TILE + (i * DW + wave) * 1024), 16, 0,
                                       0);
    }
  };
  auto stage_v = [&](int slot, int kv0) {
    if (NW > 8 && wave >= 8) return;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(v_src[i] + kv0),
                                       (lptr_t)(v_ring + slot * ATT_TILE + (i * DW + wave) * 1024), 16, 0, 0);
  };

  Frag f;
  f.row_off = l31 * 128;
  f.sw = (l31 >> 1) & 7;
  f.h2 = h2;

  f32x16 o_acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[i][e] = 0.0f;
  float m_run = -INFINITY;   // running max of raw scores (this lane's query)
  float l_run = 0.0f;        // this half-wave's share of the running sum
  const float c = p.scale_log2;
  const int n_tiles = (S + KVB - 1) / KVB;
  const bool ragged = (S & (KVB - 1)) != 0;
  // static priority for the second-dispatched half (guide T5, static form): it loses VALU arbitration to the older half on
  // every segment otherwise; one s_setprio, no per-cluster flips (the condition is provably wave-uniform: readfirstlane)
  if (NW == 8 && p.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);

  if (VARIANT == 14) {
    // Explicitly staged fragments.  hipcc's scheduler, squeezed to 128 VGPRs, sinks every ds_read_b128 next to the
    // MFMA that consumes it (variant 1's PV phase is read / wait / mfma eight times over: ~13 exposed LDS round trips
    // per tile).  Here the 8 K fragments are fetched in one batch, the first 4 V^T fragments are fetched BEFORE the
    // softmax and the last 4 while the first 4 PV MFMAs run; sched_barrier fences keep that order; the half-wave max
    // exchange is a v_permlane32_swap instead of an LDS bpermute; the ragged tail tile is peeled.
    const int n_loop = ragged ? n_tiles - 1 : n_tiles;
    stage_k(0, 0);
    stage_v(0, 0);
    auto tile = [&](int t, auto masked) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 1 < n_tiles) {
        stage_k((t + 1) & 1, (t + 1) * KVB);
        stage_v((t + 1) & 1, (t + 1) * KVB);
      }
      const char* Ks = k_ring + (t & 1) * ATT_TILE + f.row_off;
      const char* Vs = v_ring + (t & 1) * ATT_TILE + f.row_off;
      bf16x8 kf[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        kf[i] = *(const bf16x8*)(Ks + (i >> 2) * 4096 + (((2 * (i & 3) + f.h2) ^ f.sw) * 16));
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub * 4 + ks], qf[ks], s[sub], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 vf[8];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) vf[kk] = *(const bf16x8*)(Vs + (((2 * kk + f.h2) ^ f.sw) * 16));
      __builtin_amdgcn_sched_barrier(0);
      if (masked.value) mask_tail(s, t * KVB, S, h2);
      // ---- online softmax (per-lane query; the two half-waves exchange the tile max through the VALU) ----
      float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
      for (int e = 1; e < 16; ++e) mt = fmaxf(fmaxf(mt, s[0][e]), s[1][e]);
      {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
        mt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
      }
      if (__any(mt > m_run)) {
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
      }
      const float mc = m_run * c;
      const f32x2p c2 = {c, c}, mc2 = {mc, mc};
      f32x2p ps2 = {0.0f, 0.0f};
      bf16x8 pf[4];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2p sv = {s[sub][8 * g + 2 * j], s[sub][8 * g + 2 * j + 1]};
            const f32x2p x = __builtin_elementwise_fma(sv, c2, -mc2);
            const f32x2p pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            ps2 += pv;
            pk.u[j] = pack_bf2(pv.x, pv.y);
          }
          pf[sub * 2 + g] = pk.v;
        }
      l_run += ps2.x + ps2.y;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) vf[4 + kk] = *(const bf16x8*)(Vs + 4096 + (((2 * kk + f.h2) ^ f.sw) * 16));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk], pf[kk], o_acc[0], 0, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[4 + kk], pf[kk], o_acc[1], 0, 0, 0);
    };
    for (int t = 0; t < n_loop; ++t) tile(t, BoolC<false>{});
    if (ragged) tile(n_tiles - 1, BoolC<true>{});
  } else if (VARIANT == 42 || VARIANT == 43) {
    // Ping-pong (round 3): the two waves that share a SIMD (w and w + 4) run HALF A TILE APART.  A wave alternates a VALU
    // phase (softmax(t): exp2, pack, row sums) with an MFMA phase (PV(t), then QK(t + 1)); every phase boundary is a
    // workgroup barrier, group B (waves 4-7) starts one phase late, so while one wave of a SIMD feeds the matrix pipe its
    // partner runs its softmax.  In the straight loop (variant 41) the per-tile barrier lines all eight waves up in the SAME
    // phase: per wave-tile the SIMD then spends MFMA time + VALU time (measured 900 cycles against 512 of MFMA).
    //   phase 2u     (even): A = PV(u-1), QK(u)       B = softmax(u-1)        every wave issues its share of K(u+1), V(u)
    //   phase 2u + 1 (odd) : A = softmax(u)           B = PV(u-1), QK(u)      then vmcnt(0): K(u+1), V(u) have landed
    // K(t) is read in phases 2t (A) and 2t + 1 (B), V(t) in 2t + 2 and 2t + 3: K(t + 2) / V(t + 1) reuse the slots of K(t) /
    // V(t - 1) from phase 2t + 2 on -- the two-slot rings of the straight loop suffice.  43 = 42 under s_setprio (MFMA phase).
    const int T = n_tiles;
    const bool grpB = NW == 8 && wave >= 4;
    auto dma_even = [&](int u) {   // at the start of even phase 2u
      if (u + 1 < T) stage_k((u + 1) & 1, (u + 1) * KVB);
      if (u < T) stage_v(u & 1, u * KVB);
    };
    auto qk = [&](int t, f32x16 (&s)[2]) {
      qk_tile(k_ring + (t & 1) * ATT_TILE, qf, f, s);
      if (ragged && t == T - 1) mask_tail(s, t * KVB, S, h2);
    };
    f32x16 s[2];
    bf16x8 pf[4];
    stage_k(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // opens phase 0
    dma_even(0);
    if (!grpB) {
      qk(0, s);
      __syncthreads();                     // opens phase 1
      for (int t = 0; t < T; ++t) {
        softmax_tile_zero(s, m_run, l_run, o_acc, pf);            // phase 2t + 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                   // opens phase 2t + 2
        dma_even(t + 1);
        if (VARIANT == 43) __builtin_amdgcn_s_setprio(1);
        pv_tile(v_ring + (t & 1) * ATT_TILE, pf, f, o_acc);
        if (t + 1 < T) qk(t + 1, s);
        if (VARIANT == 43) __builtin_amdgcn_s_setprio(0);
        __syncthreads();                   // opens phase 2t + 3
      }
      __syncthreads();                     // group B's last phase
    } else {
      __syncthreads();                     // opens phase 1
      qk(0, s);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                     // opens phase 2
      for (int t = 0; t < T; ++t) {
        dma_even(t + 1);
        softmax_tile_zero(s, m_run, l_run, o_acc, pf);            // phase 2t + 2
        __syncthreads();                   // opens phase 2t + 3
        if (VARIANT == 43) __builtin_amdgcn_s_setprio(1);
        pv_tile(v_ring + (t & 1) * ATT_TILE, pf, f, o_acc);
        if (t + 1 < T) qk(t + 1, s);
        if (VARIANT == 43) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                   // opens phase 2t + 4
      }
    }
  } else if (VARIANT < 2 || VARIANT >= 10) {
    // VARIANT >= 16 (measurement only, wrong results, NOT reachable from the C ABI -- instantiate by hand): ablation
    // bits 1 no DMA after tile 0, 2 no LDS fragment reads, 4 no exp2, 8 no per-tile wait + barrier.  Round-1 readings
    // at the C2 shape (ms per 2-sample launch): full 8.75 | no DMA 7.62 | no LDS reads 6.76 | neither 6.10 |
    // no barrier 8.62 | no exp2 7.95 | no DMA/LDS/exp2 5.30 (MFMA floor at the sustained clock ~3.9)
    constexpr int ABL = (VARIANT >= 16 && VARIANT < 32) ? VARIANT - 16 : 0;  // 32, 33: real variants
    constexpr int NOEXP = VARIANT == 10 ? 1 : (VARIANT == 11 || (ABL & 4)) ? 2 : 0;
    // SPLIT: this workgroup owns KV tiles [t0, t1) of its unit only
    const int t0 = SPLIT ? min(chunk * p.tail_tiles, n_tiles) : 0;
    const int t1 = SPLIT ? min(t0 + p.tail_tiles, n_tiles) : n_tiles;
    if (!SPLIT || t0 < t1) {
      stage_k(t0 & 1, t0 * KVB);
      stage_v(t0 & 1, t0 * KVB);
    }
    constexpr bool PEEL = VARIANT == 13;  // the ragged last tile runs in its own copy of the body: hipcc otherwise
                                          // if-converts the tail mask into 32 v_cndmask on EVERY tile
    const int n_loop = (PEEL && ragged) ? n_tiles - 1 : t1;
    for (int t = t0; t < n_loop; ++t) {
      if (!(ABL & 8) || t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (t + 1 < t1 && !(ABL & 1)) {
        stage_k((t + 1) & 1, (t + 1) * KVB);
        stage_v((t + 1) & 1, (t + 1) * KVB);
      }
      f32x16 s[2];
      qk_tile<(ABL & 2) != 0>(k_ring + (t & 1) * ATT_TILE, qf, f, s);
      if (!PEEL && ragged && t == n_tiles - 1) mask_tail(s, t * KVB, S, h2);
      bf16x8 pf[4];
      if (VARIANT == 34 || VARIANT == 41)   // 41: Q arrives pre-scaled (alg_flash_attn_d64_ex, ALG_ATTN_Q_PRESCALED)
        softmax_tile_zero(s, m_run, l_run, o_acc, pf);
      else if (VARIANT == 36)
        softmax_tile_lazy<false>(s, c, m_run, l_run, o_acc, pf);
      else if (VARIANT == 33 || ABL != 0)   // the ablation variants (wrong results) time the default softmax
        softmax_tile_lazy(s, c, m_run, l_run, o_acc, pf);
      else
        softmax_tile<(VARIANT >= 1), NOEXP, PEEL, VARIANT == 32>(s, c, m_run, l_run, o_acc, pf);
      pv_tile<(ABL & 2) != 0>(v_ring + (t & 1) * ATT_TILE, pf, f, o_acc);
    }
    if (PEEL && ragged) {
      const int t = n_tiles - 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      f32x16 s[2];
      qk_tile(k_ring + (t & 1) * ATT_TILE, qf, f, s);
      mask_tail(s, t * KVB, S, h2);
      bf16x8 pf[4];
      softmax_tile<true, 0, true>(s, c, m_run, l_run, o_acc, pf);
      pv_tile(v_ring + (t & 1) * ATT_TILE, pf, f, o_acc);
    }
  } else {
    // K one tile ahead of V: iteration t computes S(t+1) = K(t+1) Q^T, softmax(S(t)), O += V(t)^T P(t)
    f32x16 s_a[2], s_b[2];
    stage_k(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_tiles > 1) stage_k(1, KVB);
    stage_v(0, 0);
    qk_tile(k_ring, qf, f, s_a);
    if (ragged && n_tiles == 1) mask_tail(s_a, 0, S, h2);
    int ks_next = 1;  // ring slot of K(t+1)
    // one pipeline step; called with (s_a, s_b) and (s_b, s_a) alternately so the score registers are never copied
    auto step = [&](int t, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // K(t+1) and V(t) landed; every wave is done with K(t-1)'s slot and V(t-1)'s slot
      const int ks_nn = ks_next == 2 ? 0 : ks_next + 1;
      if (t + 2 < n_tiles) stage_k(ks_nn, (t + 2) * KVB);
      if (t + 1 < n_tiles) stage_v((t + 1) & 1, (t + 1) * KVB);
      if (t + 1 < n_tiles) {
        qk_tile(k_ring + ks_next * ATT_TILE, qf, f, s_nxt);
        if (ragged && t + 1 == n_tiles - 1) mask_tail(s_nxt, (t + 1) * KVB, S, h2);
      }
      bf16x8 pf[4];
      softmax_tile<true>(s_cur, c, m_run, l_run, o_acc, pf);
      pv_tile(v_ring + (t & 1) * ATT_TILE, pf, f, o_acc);
      ks_next = ks_nn;
    };
    for (int t = 0; t < n_tiles; t += 2) {
      step(t, s_a, s_b);
      if (t + 1 < n_tiles) step(t + 1, s_b, s_a);
    }
  }

  // ---- finish: combine the half-waves' sums, normalise, store O[q][d] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (SPLIT) {  // partial result of this KV chunk: unnormalised O, running max, row sum
    const int64_t row = ((int64_t)part * p.tail_split + chunk) * (NW * 32) + wave * 32 + l31;
    float* wo = p.ws_o + row * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(wo + dt * 32 + 8 * g + 4 * h2) =
            make_float4(o_acc[dt][4 * g], o_acc[dt][4 * g + 1], o_acc[dt][4 * g + 2], o_acc[dt][4 * g + 3]);
    if (h2 == 0) *(float2*)(p.ws_ml + row * 2) = make_float2(m_run, l_tot);
    return;
  }
  const float inv = 1.0f / l_tot;
  if (q_row < S) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * h2;
        uint2 v;
        v.x = pack_bf2(o_acc[dt][4 * g] * inv, o_acc[dt][4 * g + 1] * inv);
        v.y = pack_bf2(o_acc[dt][4 * g + 2] * inv, o_acc[dt][4 * g + 3] * inv);
        *(uint2*)(op + d) = v;
      }
  }
}

// Merge of the split-KV tail: O = sum_c 2^((m_c - M) c) O_c / sum_c 2^((m_c - M) c) l_c.  One thread = 4 output values.
__global__ __launch_bounds__(256) void flash_attn_d64_merge_kernel(const AttnP p) {
  constexpr int QBR = 256;  // query rows per unit (8 waves x 32)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)8 * p.tail_units * QBR * 16;
  if (e >= total) return;
  const int d4 = (int)(e & 15);
  const int row = (int)((e >> 4) % QBR);
  const int part = (int)(e / (16 * QBR));
  const int xcd = part / p.tail_units, u = part - xcd * p.tail_units;
  const int idx = p.unit0 + u;
  const int slot = idx / p.q_blocks, qb = idx - slot * p.q_blocks;
  const int bh = slot * 8 + xcd;
  const int q_row = qb * QBR + row;
  if (bh >= p.batch * p.heads || q_row >= p.S) return;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int64_t base = (int64_t)part * p.tail_split * QBR + row;
  float M = -INFINITY;
  for (int c = 0; c < p.tail_split; ++c) M = fmaxf(M, p.ws_ml[(base + (int64_t)c * QBR) * 2]);
  float L = 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < p.tail_split; ++c) {
    const int64_t r = base + (int64_t)c * QBR;
    const float2 ml = *(const float2*)(p.ws_ml + r * 2);
    if (ml.y == 0.0f) continue;  // empty chunk
    const float w = __builtin_amdgcn_exp2f((ml.x - M) * p.scale_log2);
    const float4 o = *(const float4*)(p.ws_o + r * 64 + d4 * 4);
    L += w * ml.y;
    acc.x += w * o.x, acc.y += w * o.y, acc.z += w * o.z, acc.w += w * o.w;
  }
  const float inv = 1.0f / L;
  uint2 v;
  v.x = pack_bf2(acc.x * inv, acc.y * inv);
  v.y = pack_bf2(acc.z * inv, acc.w * inv);
  *(uint2*)(p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 64 + d4 * 4) = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined form (round 3, ALG_ATTN_PP=3): the steady-state KV loop is ONE generated asm statement (attn_pipe_loop.inc,
// scripts/gen_attn_pipe.py) in which every MFMA is followed, in program order of the SAME wave, by the softmax work of one
// score pair and one fragment read -- the only arrangement in which matrix and vector work overlap on this part
// (profiles/r3_attention_d64_mix_microbench.txt).  This kernel is the frame around it: the same workgroup -> (head, q block)
// map, Q / K / V^T layouts and finish as flash_attn_d64_kernel<41>, a C++ loop that runs tile 0 (where the running offset is
// established and snapped to zero), the last few tiles (ragged tail) and any tile on which the statement bails out (row sum
// outside [0, 2^80): the exact max / rescale path), all under the statement's collective protocol:
//     top of iteration t:  s_waitcnt vmcnt(0); s_barrier; DMA K(t+2) -> K slot (t+2) & 3, V^T(t+1) -> V slot (t+1) & 3
// so that the waves of a workgroup may be inside or outside the statement independently.  The straight form of an iteration
// reads K(t), V^T(t); the pipelined form K(t+1), V^T(t-1): four-slot rings keep all of them resident.
// Row sums are plain fp32 adds of the unrounded probabilities inside the statement (v_dot2c does not hide behind an MFMA),
// the bf16-rounded dot2 sums of softmax_tile_zero outside it.
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void flash_attn_d64_pipe_kernel(const AttnP p) {
  // NW = 4 (default): four waves x 32 queries, the statement names v[64:165] and a[0:79] (252 registers), two of these
  // workgroups (2 x 64 KiB of LDS) share a CU.  NW = 8 (ALG_ATTN_PP=4): one 256-query unit per workgroup, half the L2 -> LDS
  // traffic per MFMA; hipcc grants an 8-wave workgroup 128 + 128 registers per lane, so that form of the statement lives in
  // v[26:127] and takes O in AccVGPR operands.
  constexpr int ROUNDS = 8 / NW;
  __shared__ __attribute__((aligned(16))) char smem[8 * ATT_TILE];
  char* const k_ring = smem;
  char* const v_ring = smem + 4 * ATT_TILE;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nbh = p.batch * p.heads;
  // p.q_blocks counts 256-query units (the unit of flash_attn_d64_kernel and of the split-KV tail plan); a unit is two of this
  // kernel's 128-query workgroups, neighbours in the grid
  int bh, qb;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    const int unit = NW == 4 ? idx >> 1 : idx;
    const int slot = unit / p.q_blocks;
    qb = NW == 4 ? (unit - slot * p.q_blocks) * 2 + (idx & 1) : unit - slot * p.q_blocks;
    bh = slot * 8 + xcd;
    if (bh >= nbh) return;
  }
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int S = p.S;
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* K = p.k + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 64 * p.vt_rs;
  const int T = (S + KVB - 1) / KVB;
  const bool ragged = (S & (KVB - 1)) != 0;
  // clock tap (bench.py: the shader clock THIS kernel ran at): scalar reads of two counters, wave 0 of every 64th workgroup
  const bool tap = p.clk != nullptr && (blockIdx.x & 63) == 0 && (int)(blockIdx.x >> 6) < p.clk_slots && wave == 0;
  uint64_t tap_c0 = 0, tap_r0 = 0;
  if (tap) {
    tap_c0 = __builtin_readcyclecounter();
    tap_r0 = wall_clock64();
  }
  f32x16 oa[2];
#pragma unroll
  for (int i = 0; i < 32; ++i) oa[i >> 4][i & 15] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  // Everything lane-derived is rebuilt from a lane id (LaneCtx): the C++ loops in front of and behind the statement each build
  // their own from a freshly laundered id, so that none of it is live across the statement (values that are compete with its
  // 32 O operands for v[0:63], and hipcc then parks them in AccVGPRs beyond a79: 328 registers, one wave per SIMD)
  struct LaneCtx {
    int lane, l31, h2, tid, srow, sslot, q_row;
    Frag f;
  };
  auto make_ctx = [&](int lane) -> LaneCtx {
    LaneCtx c;
    c.lane = lane, c.l31 = lane & 31, c.h2 = lane >> 5, c.tid = wave * 64 + lane;
    c.srow = c.tid >> 3, c.sslot = (c.tid & 7) ^ ((c.tid >> 4) & 7);
    c.q_row = qb * (NW * 32) + wave * 32 + c.l31;
    c.f.row_off = c.l31 * 128, c.f.sw = (c.l31 >> 1) & 7, c.f.h2 = c.h2;
    return c;
  };
  auto fresh_lane = [&]() -> int {
    int z = 0;
    asm volatile("" : "+s"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  };
  auto stage_k = [&](const LaneCtx& c, int t) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      const bf16_t* ks = K + (int64_t)min(t * KVB + c.srow + NW * 8 * i, S - 1) * p.q_rs + c.sslot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + (t & 3) * ATT_TILE + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](const LaneCtx& c, int t) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(VT + (int64_t)(c.srow + NW * 8 * i) * p.vt_rs + c.sslot * 8 + min(t, T - 1) * KVB),
                                       (lptr_t)(v_ring + (t & 3) * ATT_TILE + (i * NW + wave) * 1024), 16, 0, 0);
  };
  // iterations [t, t_end) in the straight form: protocol (unless the first one's is already done), QK(t) -> softmax -> PV(t)
  auto straight = [&](const LaneCtx& c, int t, int t_end, bool top_done) {
    bf16x8 qf[4];
    const bf16_t* qp = Q + (int64_t)min(c.q_row, S - 1) * p.q_rs + c.h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    for (; t < t_end; ++t) {
      if (!top_done) {
        if constexpr (ROUNDS == 2)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the previous iteration's DMAs (2 * ROUNDS per wave)
        else
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
        stage_k(c, t + 3);   // past the end the source rows are clamped (K) / lie in the padded pitch (V^T): the DMA count per
        stage_v(c, t + 2);   // iteration must not depend on t, the counted wait above relies on it
      }
      top_done = false;
      f32x16 s[2];
      qk_tile(k_ring + (t & 3) * ATT_TILE, qf, c.f, s);
      if (ragged && t == T - 1) mask_tail(s, t * KVB, S, c.h2);
      bf16x8 pf[4];
      softmax_tile_zero(s, m_run, l_run, oa, pf);
      pv_tile(v_ring + (t & 3) * ATT_TILE, pf, c.f, oa);
    }
  };

  // the statement only runs iterations t whose DMA target K(t + 3) is a whole tile (its sources are not clamped) and whose
  // tile t + 1 needs no mask
  const int tend = ragged ? T - 4 : T - 3;
  int t = 1;
  bool top_done = false;
  {
    const LaneCtx c = make_ctx(fresh_lane());
    stage_k(c, 0);
    stage_k(c, 1);
    stage_v(c, 0);
    stage_v(c, 0);       // (filler: four DMAs per batch)
    stage_k(c, 2);       // the batch "iteration -1" would have issued: K(2), V(1)
    stage_v(c, 1);
    straight(c, 0, 1, false);     // tile 0: establishes the running offset (snapped to zero when its scores allow)
  }
  if (1 + 4 <= tend && __all(m_run == 0.0f)) {
    const LaneCtx c = make_ctx(fresh_lane());
    auto sreg = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
    auto uniform64 = [](const void* ptr) -> uint64_t {
      const uint64_t v = (uint64_t)(uintptr_t)ptr;
      return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint32_t kl = (uint32_t)(uintptr_t)(lptr_t)k_ring, vl = (uint32_t)(uintptr_t)(lptr_t)v_ring;
    const int fl0 = c.f.row_off + (((0 + c.h2) ^ c.f.sw) * 16), fl1 = c.f.row_off + (((2 + c.h2) ^ c.f.sw) * 16);
    const int fl2 = c.f.row_off + (((4 + c.h2) ^ c.f.sw) * 16), fl3 = c.f.row_off + (((6 + c.h2) ^ c.f.sw) * 16);
    const int lk0 = kl + fl0, lk1 = kl + fl1, lk2 = kl + fl2, lk3 = kl + fl3;
    const int lv0 = vl + fl0, lv1 = vl + fl1, lv2 = vl + fl2, lv3 = vl + fl3;
    int kvo0 = (int)(((int64_t)((t + 3) * KVB + c.srow) * p.q_rs + c.sslot * 8) * 2);
    int vvo0 = (int)(((int64_t)c.srow * p.vt_rs + c.sslot * 8 + (t + 2) * KVB) * 2);
    const int qvo = (int)(((int64_t)min(c.q_row, S - 1) * p.q_rs + c.h2 * 8) * 2);
    const uint64_t kb = uniform64(K), vb = uniform64(VT), qbs = uniform64(Q);
    const int kstep = sreg((int)(KVB * p.q_rs * 2)), tend_s = sreg(tend);
    const int wk = sreg((int)kl + wave * 1024), wv = sreg((int)vl + wave * 1024);
    int ts = sreg(t), code;
    float o[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i] = oa[i >> 4][i & 15];
    if constexpr (NW == 4) {
      int kvo1 = (int)(((int64_t)((t + 3) * KVB + c.srow + 32) * p.q_rs + c.sslot * 8) * 2);
      int vvo1 = (int)(((int64_t)(c.srow + 32) * p.vt_rs + c.sslot * 8 + (t + 2) * KVB) * 2);
      asm volatile(ALG_ATTN_PIPE_LOOP_ASM
                   : ALG_ATTN_PIPE_O_OPERANDS(o), [l] "+v"(l_run), [t] "+s"(ts), [code] "=&s"(code), [kvo0] "+v"(kvo0),
                     [kvo1] "+v"(kvo1), [vvo0] "+v"(vvo0), [vvo1] "+v"(vvo1)
                   : [lk0] "v"(lk0), [lk1] "v"(lk1), [lk2] "v"(lk2), [lk3] "v"(lk3), [lv0] "v"(lv0), [lv1] "v"(lv1),
                     [lv2] "v"(lv2), [lv3] "v"(lv3), [qvo] "v"(qvo), [kb] "s"(kb), [vb] "s"(vb), [qb] "s"(qbs),
                     [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                   : "memory", "vcc", "scc", ALG_ATTN_PIPE_CLOBBERS);
    } else {
      asm volatile(ALG_ATTN_PIPE8_LOOP_ASM
                   : ALG_ATTN_PIPE8_O_OPERANDS(o), [l] "+v"(l_run), [t] "+s"(ts), [code] "=&s"(code), [kvo0] "+v"(kvo0),
                     [vvo0] "+v"(vvo0)
                   : [lk0] "v"(lk0), [lk1] "v"(lk1), [lk2] "v"(lk2), [lk3] "v"(lk3), [lv0] "v"(lv0), [lv1] "v"(lv1),
                     [lv2] "v"(lv2), [lv3] "v"(lv3), [qvo] "v"(qvo), [kb] "s"(kb), [vb] "s"(vb), [qb] "s"(qbs),
                     [kstep] "s"(kstep), [tend] "s"(tend_s), [wk] "s"(wk), [wv] "s"(wv)
                   : "memory", "vcc", "scc", ALG_ATTN_PIPE8_CLOBBERS);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) oa[i >> 4][i & 15] = o[i];
    t = ts;
    top_done = code != 0;   // 1: iteration t's protocol is done, softmax(t) is not: tile t is redone below
  }
  LaneCtx c = make_ctx(fresh_lane());
  straight(c, t, T, top_done);   // the tiles behind the statement (or all of them but tile 0)

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (c.q_row < S) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)c.q_row * p.o_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * c.h2;
        uint2 v;
        v.x = pack_bf2(oa[dt][4 * g] * inv, oa[dt][4 * g + 1] * inv);
        v.y = pack_bf2(oa[dt][4 * g + 2] * inv, oa[dt][4 * g + 3] * inv);
        *(uint2*)(op + d) = v;
      }
  }
  if (tap && c.lane == 0) {
    uint64_t* cp = p.clk + (size_t)(blockIdx.x >> 6) * 4;   // one workgroup owns a slot (block / 64 < slots)
    cp[0] = tap_c0, cp[1] = tap_r0, cp[2] = __builtin_readcyclecounter(), cp[3] = wall_clock64();
  }
}


// Workgroup-count quantisation (measured, scripts/attn_tail_probe.py): every XCD runs 64 workgroups at a time (32 CUs x
// 2), a workgroup takes ~0.64 ms at S = 17,776, and a 2-sample C2 launch is 840 units per XCD = 13.125 rounds: the
// fourteenth round keeps 8 of 64 slots busy and costs 1.5-5 % of the launch depending on the box.  When the last round is at most 1/4 full its units
// are cut along KV instead: tail_units x split chunk-workgroups fill the slots, then one small merge kernel.
struct TailPlan {
  int units, split, tiles;
};
static TailPlan plan_tail(int nbh, int q_blocks, int n_tiles) {
  TailPlan t = {0, 0, 0};
  if (!opt(OPT_ATTN_SPLIT_TAIL)) return t;  // ALG_ATTN_SPLIT_TAIL=0: single launch (parity tests: batch-shape bit-identity)
  if (nbh % 8) return t;  // heads spread unevenly over the XCDs: a different imbalance, not this one
  const int slots = 64;
  const int per_xcd = nbh / 8 * q_blocks;
  const int r = per_xcd % slots;
  if (per_xcd < slots || r == 0 || r > 16 || n_tiles < 64) return t;  // fuller last rounds gain little
  t.units = r;
  t.split = (r * 8) % slots == 0 ? 8 : 16;
  t.tiles = (n_tiles + t.split - 1) / t.split;
  return t;
}

// 33 = the default (dot2 row sums + lazy running max); 1 = the exact-running-max reference (fp32 row sums) the parity tests
// compare it with (capi.hip accepts only {1, 33}).
static int attn_variant() {
  const int v = opt(OPT_ATTN_VARIANT);
  return v == 1 ? 1 : 33;
}

}  // namespace alg

using namespace alg;

// Main launch of the pre-scaled form: 4 = the pipelined kernel, one 8-wave workgroup per 256-query unit (default since round 3:
// asm steady-state loop, every MFMA followed by one score pair of the softmax), 0 = the straight loop (variant 41; same
// softmax, fp32 summation order differs); 7 = attention64_m16.hip (the statement on 16x16x32 MFMAs), taken by
// flash_attn_d64_m16() before this function is reached.
namespace alg {
// attention64_m16.hip: the 8-wave statement kernel on v_mfma_f32_16x16x32_bf16 as the main launch (ALG_ATTN_PP=7); 1 = not covered
int flash_attn_d64_m16(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S, int q_blocks,
                       int64_t q_bs, int64_t q_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs, int64_t o_rs, unsigned blocks,
                       hipStream_t stream);
// the main launch by another file (ALG_ATTN_PP = 7), or 1: attention.hip's own kernels take the call
static int main_launch_elsewhere(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S, int q_blocks,
                                 int64_t q_bs, int64_t q_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs, int64_t o_rs,
                                 unsigned blocks, hipStream_t stream) {
  const int pp = opt(OPT_ATTN_PP);
  if (pp == 7) return flash_attn_d64_m16(q, k, vt, o, batch, heads, S, q_blocks, q_bs, q_rs, vt_bs, vt_rs, o_bs, o_rs, blocks, stream);
  return 1;
}
}
static void launch_main41(dim3 g, dim3 blk, hipStream_t s, const alg::AttnP& p) {
  int pp = opt(OPT_ATTN_PP);
  // the pipelined statements address K / V^T / Q with 31-bit byte offsets from the (batch, head) panel bases
  if (pp >= 3 && ((int64_t)(p.S + 4 * alg::KVB) * p.q_rs * 2 >= (1ll << 31) || (int64_t)65 * p.vt_rs * 2 >= (1ll << 31))) pp = 0;
  switch (pp) {
    case 4:   // the default: one 8-wave workgroup per unit
    case 7:   // a call flash_attn_d64_m16() declined (fewer than 12 KV tiles, 31-bit offsets, V^T pitch): the 32x32x16 statement kernel
      hipLaunchKernelGGL(alg::flash_attn_d64_pipe_kernel<8>, g, dim3(512), 0, s, p);
      break;
    default: hipLaunchKernelGGL((alg::flash_attn_d64_kernel<41, 8>), g, blk, 0, s, p); break;
  }
}

extern "C" int alg_flash_attn_d64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S,
                                  int64_t q_bstride, int64_t q_rstride, int64_t vt_bstride, int64_t vt_rstride,
                                  int64_t o_bstride, int64_t o_rstride, float scale, void* stream) {
  return alg_flash_attn_d64_ex(q, k, vt, o, batch, heads, S, q_bstride, q_rstride, vt_bstride, vt_rstride, o_bstride,
                               o_rstride, scale, 0, nullptr, 0, stream);
}

// Bytes of the split-KV workspace the launch plan of (batch, heads, S) uses (0: the plan is a single launch).  The library
// never allocates: the caller owns the buffer and passes it to alg_flash_attn_d64_ex.
extern "C" int64_t alg_flash_attn_d64_workspace_bytes(int batch, int heads, int S, int flags) {
  if (batch <= 0 || heads <= 0 || S <= 0) return 0;
  int variant = attn_variant();
  if (flags & ALG_ATTN_Q_PRESCALED) variant = 41;
  if (variant != 33 && variant != 41) return 0;
  const int q_blocks = (S + 8 * 32 - 1) / (8 * 32);
  const TailPlan tp = plan_tail(batch * heads, q_blocks, (S + KVB - 1) / KVB);
  if (!tp.units) return 0;
  return (int64_t)8 * tp.units * tp.split * 256 * 66 * (int64_t)sizeof(float);
}

extern "C" int alg_flash_attn_d64_ex(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S,
                                     int64_t q_bstride, int64_t q_rstride, int64_t vt_bstride, int64_t vt_rstride,
                                     int64_t o_bstride, int64_t o_rstride, float scale, int flags, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (!q || !k || !vt || !o || batch <= 0 || heads <= 0 || S <= 0) {
    set_error("alg_flash_attn_d64: bad argument (batch=%d heads=%d S=%d)", batch, heads, S);
    return ALG_EINVAL;
  }
  if (q_rstride % 8 || q_bstride % 8 || vt_rstride % 8 || vt_bstride % 8 || o_rstride % 4 || o_bstride % 4 ||
      ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)o & 7)) {
    set_error("alg_flash_attn_d64: q/k/vt need 16-byte aligned rows (strides %% 8 == 0), o 8-byte aligned");
    return ALG_EINVAL;
  }
  if (vt_rstride < (int64_t)((S + KVB - 1) / KVB) * KVB) {
    set_error("alg_flash_attn_d64: vt row stride %lld must cover S rounded up to %d", (long long)vt_rstride, KVB);
    return ALG_EINVAL;
  }
  const bool vt128 = vt_rstride >= (int64_t)((S + 127) / 128) * 128;  // the 128-kv stage variants read one tile further
  AttnP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.S = S;
  int variant = attn_variant();
  if (variant >= 8 && variant < 32 && !vt128) variant = 33;
  if (flags & ALG_ATTN_Q_PRESCALED) variant = 41;  // the only kernel that takes log2-unit scores
  const int nw = (variant == 5 || variant == 7 || variant == 16 || variant == 40) ? 4 : (variant == 12 ? 16 : 8);
  const int q_per_wave = (variant == 6 || variant == 7 || variant == 15 || variant == 16 || variant == 39 || variant == 40) ? 64 : 32;
  p.q_blocks = (S + nw * q_per_wave - 1) / (nw * q_per_wave);
  p.q_bs = q_bstride; p.q_rs = q_rstride; p.vt_bs = vt_bstride; p.vt_rs = vt_rstride;
  p.o_bs = o_bstride; p.o_rs = o_rstride;
  p.scale_log2 = (flags & ALG_ATTN_Q_PRESCALED) ? 1.0f : scale * 1.4426950408889634f;  // m is in log2 units already
  p.prio = 0;
  p.clk = clock_tap_for((hipStream_t)stream, &p.clk_slots);
  const int nbh = batch * heads;
  const int64_t grid = (int64_t)((nbh + 7) / 8) * 8 * p.q_blocks;
  const dim3 blk(nw * 64);
  hipStream_t s = (hipStream_t)stream;
  p.unit0 = p.tail_units = p.tail_split = p.tail_tiles = 0;
  p.ws_o = p.ws_ml = nullptr;
  if (variant == 33 || variant == 41) {
    const TailPlan tp = plan_tail(nbh, p.q_blocks, (S + KVB - 1) / KVB);
    const size_t rows = (size_t)8 * tp.units * tp.split * 256;
    // the split-KV tail runs only in a caller-provided workspace (alg_flash_attn_d64_workspace_bytes); without one the
    // whole problem is the single launch below (same rows up to fp32 summation order in the tail units)
    if (tp.units && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (int64_t)(rows * 66 * sizeof(float))) {
      const int per_xcd = nbh / 8 * p.q_blocks;
      p.unit0 = per_xcd - tp.units, p.tail_units = tp.units, p.tail_split = tp.split, p.tail_tiles = tp.tiles;
      float* ws = (float*)workspace;
      p.ws_o = ws, p.ws_ml = ws + rows * 64;
      if (variant == 41) {
        const int rq = p.unit0 > 0 ? main_launch_elsewhere(q, k, vt, o, batch, heads, S, p.q_blocks, q_bstride, q_rstride, vt_bstride,
                                                           vt_rstride, o_bstride, o_rstride, (unsigned)(8 * p.unit0), s)
                                   : 0;
        if (rq < 0 || rq > 1) return rq;
        if (rq == 1) launch_main41(dim3((unsigned)(8 * p.unit0)), blk, s, p);
        hipLaunchKernelGGL((flash_attn_d64_kernel<41, 8, true>), dim3((unsigned)(8 * tp.units * tp.split)), blk, 0, s, p);
      } else {
        hipLaunchKernelGGL((flash_attn_d64_kernel<33, 8>), dim3((unsigned)(8 * p.unit0)), blk, 0, s, p);
        hipLaunchKernelGGL((flash_attn_d64_kernel<33, 8, true>), dim3((unsigned)(8 * tp.units * tp.split)), blk, 0, s, p);
      }
      const int64_t merge = (int64_t)8 * tp.units * 256 * 16;
      hipLaunchKernelGGL(flash_attn_d64_merge_kernel, dim3((unsigned)((merge + 255) / 256)), dim3(256), 0, s, p);
      return check_launch("alg_flash_attn_d64");
    }
  }
  const dim3 g((unsigned)grid);
  if (variant == 41) {
    const int rq = main_launch_elsewhere(q, k, vt, o, batch, heads, S, p.q_blocks, q_bstride, q_rstride, vt_bstride, vt_rstride,
                                         o_bstride, o_rstride, (unsigned)grid, s);
    if (rq != 1) return rq;
  }
  switch (variant) {
    case 1: hipLaunchKernelGGL(flash_attn_d64_kernel<1>, g, blk, 0, s, p); break;
    case 41: launch_main41(g, blk, s, p); break;
    default: hipLaunchKernelGGL((flash_attn_d64_kernel<33, 8>), g, blk, 0, s, p); break;
  }
  return check_launch("alg_flash_attn_d64");
}
