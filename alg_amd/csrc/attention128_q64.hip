// Flash attention forward, head_dim 128, the LONG self-attention form: 64 queries per wave, one wave per SIMD.
//
// Why a second d = 128 kernel.  attention128.hip gives every wave 32 queries, so each of a workgroup's 8 waves pulls the
// whole 32 KiB K / V^T tile out of LDS for its 32 MFMAs: 256 KiB of fragment reads per 64-key tile and workgroup, i.e.
// 128 B/clk/CU at full MFMA rate -- exactly the LDS port (the kernel sits at 54 % of the matrix peak at its clock).  With
// 64 queries per wave every K / V^T fragment feeds TWO MFMAs (the two 32-query halves), the fragment traffic per FLOP
// halves (78 B/clk with the DMA writes), and at d = 128 the softmax of 64 x 64 scores (~192 VALU instructions) fits in the
// issue shadow of the tile's 64 MFMAs (3 per MFMA).  The price is registers: Q 64 + S 2 x 64 + P 2 x 32 + O 128 -> one wave
// per SIMD (launch bounds 256), so nothing hides a stall but the wave's own instruction stream.  Hence a software pipeline
// over KV tiles, one scheduling region per tile:
//
//     region(u) = { S_next = K(u+1) Q^T          (16 MFMA, 8 K-fragment reads)         u = a 32-key half of a 64-key tile
//                   O     += V^T(u-1) P_prev^T   (16 MFMA, 8 V-fragment reads)
//                   P_cur  = softmax numerators of S_cur = half-tile u   (VALU: fma, exp2, pack, dot2 row sums) }
//
// interleaved MFMA : ds_read : VALU = 1 : 0.5 : 3 with sched_group_barrier; the lazy running max of attention128.hip keeps
// the exact rescale path out of the region (it runs after it, rarely).  K and V^T tiles stream through two 4-slot LDS rings
// (2 x 64 KiB, 16-byte global_load_lds) three tiles ahead: per 64-key tile ONE counted wait (vmcnt(16)) and ONE barrier.
//
// Used for non-causal, ungrouped attention over at least MIN_TILES KV tiles (the Wan / HunyuanVideo self-attention); the
// cross-attentions, the causal grouped-query form and short sequences stay on attention128.hip.  Same operand layout, same
// swizzles, same accumulation order per query as that kernel (S^T = K Q^T, P as the B operand of O^T = V^T P^T).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace alg {
namespace a128q {

constexpr int NW = 4;
constexpr int QW = 64;                   // queries per wave
constexpr int KVB = 64;
constexpr int K_TILE = KVB * 128 * 2;    // 16 KiB
constexpr int V_TILE = 128 * KVB * 2;    // 16 KiB
constexpr int NS = 4;                    // ring slots per operand
constexpr int LDS_BYTES = NS * (K_TILE + V_TILE);
constexpr int MIN_TILES = 8;

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

// The K / V^T fragment ring (four 16-byte fragments, three reads in flight at any time, ACROSS regions, branches and the loop
// back edge) lives in a[240:255] and exists only inside asm text: a C++ value that an asm ds_read "returns" may be copied by
// the compiler (phi copies at a join, live-range splits) before the data has arrived -- it does not know the asm is a load.
// Every asm that touches the ring lists all sixteen registers as clobbers, so hipcc keeps its own (long-lived) AccVGPR
// values out of them.  MFMA operands may be AccVGPRs (srcA the fragment, srcB the Q fragment), ds_read can target them.
#define ALG_FRAG_CLOBBER "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", \
                         "a252", "a253", "a254", "a255"
#define ALG_FR0 "a[240:243]"
#define ALG_FR1 "a[244:247]"
#define ALG_FR2 "a[248:251]"
#define ALG_FR3 "a[252:255]"
template <int SLOT, int OFF>
__device__ __forceinline__ void frag_read(uint32_t addr) {
#ifdef ALG_Q64_NO_READS   // build-time ablations (garbage results; timing shows what the loop is bound by)
  return;
#endif
  if constexpr (SLOT == 0) asm volatile("ds_read_b128 " ALG_FR0 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 1) asm volatile("ds_read_b128 " ALG_FR1 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 2) asm volatile("ds_read_b128 " ALG_FR2 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 3) asm volatile("ds_read_b128 " ALG_FR3 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
}
// S (ArchVGPRs) = / += fragment[SLOT] x Q fragment (AccVGPRs)
template <int SLOT, bool FIRST>
__device__ __forceinline__ void qk_mfma(f32x16& s, const bf16x8 qv) {
#define ALG_QK(FR)                                                                                                     \
  if constexpr (FIRST)                                                                                                 \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, " FR ", %1, 0" : "=v"(s) : "a"(qv) : ALG_FRAG_CLOBBER);               \
  else                                                                                                                 \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, " FR ", %1, %0" : "+v"(s) : "a"(qv) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 0) { ALG_QK(ALG_FR0) }
  if constexpr (SLOT == 1) { ALG_QK(ALG_FR1) }
  if constexpr (SLOT == 2) { ALG_QK(ALG_FR2) }
  if constexpr (SLOT == 3) { ALG_QK(ALG_FR3) }
#undef ALG_QK
}
// O lives in FIXED AccVGPRs: tile (qh, dt) = a[16 (4 qh + dt) .. + 15].  Every MFMA that accumulates into O names them, and
// the (rare) exact-rescale path multiplies them in place inside one asm block -- otherwise hipcc keeps the O tiles that the
// cold path touches in ArchVGPRs and copies them out of the AccVGPRs on the HOT path (64 v_accvgpr_read per region, issued
// right behind MFMAs it cannot see inside the asm).
template <int IDX, int SLOT>
__device__ __forceinline__ void pv_mfma(f32x16& o, const bf16x8 pfrag) {
#define ALG_PV2(LO, HI, FR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, " FR ", %1, %0" : "+{a[" #LO ":" #HI "]}"(o) : "v"(pfrag) : ALG_FRAG_CLOBBER);
#define ALG_PV_CASE(I, LO, HI)                        \
  if constexpr (IDX == I) {                           \
    if constexpr (SLOT == 0) { ALG_PV2(LO, HI, ALG_FR0) } \
    if constexpr (SLOT == 1) { ALG_PV2(LO, HI, ALG_FR1) } \
    if constexpr (SLOT == 2) { ALG_PV2(LO, HI, ALG_FR2) } \
    if constexpr (SLOT == 3) { ALG_PV2(LO, HI, ALG_FR3) } \
  }
  ALG_PV_CASE(0, 0, 15) ALG_PV_CASE(1, 16, 31) ALG_PV_CASE(2, 32, 47) ALG_PV_CASE(3, 48, 63)
  ALG_PV_CASE(4, 64, 79) ALG_PV_CASE(5, 80, 95) ALG_PV_CASE(6, 96, 111) ALG_PV_CASE(7, 112, 127)
#undef ALG_PV_CASE
#undef ALG_PV2
}
#define ALG_RS1(R) "v_accvgpr_read_b32 %4, a" #R "\n\tv_mul_f32 %4, %4, %5\n\tv_accvgpr_write_b32 a" #R ", %4\n\t"
#define ALG_RS16(B) ALG_RS1(B##0) ALG_RS1(B##1) ALG_RS1(B##2) ALG_RS1(B##3) ALG_RS1(B##4) ALG_RS1(B##5) ALG_RS1(B##6) ALG_RS1(B##7) ALG_RS1(B##8) ALG_RS1(B##9)
// O[qh] *= alpha, all four d-tiles, in the AccVGPRs they are pinned to (two leading s_nop 15: the region's last MFMAs wrote
// O a few cycles ago, and hipcc does not see an MFMA inside an asm)
__device__ __forceinline__ void rescale_o(f32x16 (&o)[4], float alpha, int qh) {
  float tmp;
  if (qh == 0) {
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
                 ALG_RS16() ALG_RS16(1) ALG_RS16(2) ALG_RS16(3) ALG_RS16(4) ALG_RS16(5)
                 ALG_RS1(60) ALG_RS1(61) ALG_RS1(62) ALG_RS1(63)
                 : "+{a[0:15]}"(o[0]), "+{a[16:31]}"(o[1]), "+{a[32:47]}"(o[2]), "+{a[48:63]}"(o[3]), "=&v"(tmp)
                 : "v"(alpha));
  } else {
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
                 ALG_RS1(64) ALG_RS1(65) ALG_RS1(66) ALG_RS1(67) ALG_RS1(68) ALG_RS1(69)
                 ALG_RS16(7) ALG_RS16(8) ALG_RS16(9) ALG_RS16(10) ALG_RS16(11)
                 ALG_RS1(120) ALG_RS1(121) ALG_RS1(122) ALG_RS1(123) ALG_RS1(124) ALG_RS1(125) ALG_RS1(126) ALG_RS1(127)
                 : "+{a[64:79]}"(o[0]), "+{a[80:95]}"(o[1]), "+{a[96:111]}"(o[2]), "+{a[112:127]}"(o[3]), "=&v"(tmp)
                 : "v"(alpha));
  }
}
#undef ALG_RS16
#undef ALG_RS1

__global__ __launch_bounds__(NW * 64) void flash_attn_d128_q64_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const k_ring = smem;
  char* const v_ring = smem + NS * K_TILE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h2 = lane >> 5;

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

  // Q^T fragments (B operand) of the wave's two 32-query halves: lane (q = l31, h2) holds Q[q][16 ks + 8 h2 .. +8]
  const int q_row0 = qb * (NW * QW) + wave * QW + l31;
  bf16x8 qf[2][8];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const bf16_t* qp = Q + (int64_t)min(q_row0 + qh * 32, Sq - 1) * p.q_rs + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qh][ks] = *(const bf16x8*)(qp + ks * 16);
  }

  const int n_tiles = (Skv + KVB - 1) / KVB;
  const bool ragged = (Skv & (KVB - 1)) != 0;

  // DMA: buffer_load ... lds with the tile's origin in the SCALAR offset and a per-lane byte offset that never changes -- no
  // vector arithmetic per tile (the global_load_lds form cost ~5 VALU instructions per DMA for its 64-bit lane addresses, in
  // a loop whose vector pipe is the busy one), and rows past Skv read as zeros through the descriptor's bounds check instead
  // of a per-lane clamp.  K tile: 64 rows x 16 slots (256 B rows), four rounds of 16 rows; physical slot tid & 15 holds
  // logical slot (tid & 15) ^ (row & 15).  V^T tile: 128 d-rows x 8 slots, four rounds of 32 rows, swizzle (row >> 1) & 7.
  // Tiles past the end re-fetch the last one (uniform instruction counts for the counted waits; nobody uses the data).
  const int k_rs = (int)p.k_rs, vt_rs = (int)p.vt_rs;
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)(((int64_t)(Skv - 1) * k_rs + 128) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)VT, 0, (int)((int64_t)128 * vt_rs * 2), 0x00020000);
  int k_vo[4], v_vo[4];
  {
    const int row = tid >> 4, slot = (tid & 15) ^ ((tid >> 4) & 15);
    const int vrow = tid >> 3, vslot = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      k_vo[i] = ((row + i * 16) * k_rs + slot * 8) * 2;
      v_vo[i] = ((vrow + i * 32) * vt_rs + vslot * 8) * 2;
    }
  }
  // one DMA instruction of the pair [K(tk), V(tv)]: pieces 0 - 3 the K rounds, 4 - 7 the V^T rounds
  auto stage_piece = [&](int tk, int tv, auto piece_c) {
    constexpr int PC = decltype(piece_c)::value;
    if constexpr (PC < 4) {
      const int so = min(tk, n_tiles - 1) * KVB * k_rs * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(k_ring + (tk & (NS - 1)) * K_TILE + (PC * 256 + wave * 64) * 16), 16,
                                               k_vo[PC], so, 0, 0);
    } else {
      const int so = min(tv, n_tiles - 1) * KVB * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(v_ring + (tv & (NS - 1)) * V_TILE + ((PC - 4) * 256 + wave * 64) * 16),
                                               16, v_vo[PC - 4], so, 0, 0);
    }
  };
  auto stage_k = [&](int tile) {
    stage_piece(tile, 0, std::integral_constant<int, 0>{});
    stage_piece(tile, 0, std::integral_constant<int, 1>{});
    stage_piece(tile, 0, std::integral_constant<int, 2>{});
    stage_piece(tile, 0, std::integral_constant<int, 3>{});
  };
  auto stage_v = [&](int tile) {
    stage_piece(0, tile, std::integral_constant<int, 4>{});
    stage_piece(0, tile, std::integral_constant<int, 5>{});
    stage_piece(0, tile, std::integral_constant<int, 6>{});
    stage_piece(0, tile, std::integral_constant<int, 7>{});
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const int k_row_off = l31 * 256, k_sw = l31 & 15;
  const int v_row_off = l31 * 128, v_sw = (l31 >> 1) & 7;
  const float c = p.scale_log2;
  // per-lane fragment addresses without the (slot, half, d-tile) part: kc[k-step], vc[32-key block of the 64-key tile]
  uint32_t kc[8], vc[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kc[ks] = lds0 + k_row_off + (((2 * ks + h2) ^ k_sw) * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) vc[j] = lds0 + NS * K_TILE + v_row_off + (((2 * j + h2) ^ v_sw) * 16);

  f32x16 o_acc[2][4];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[qh][i][e] = 0.0f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};

  // The KV tile is STAGED 64 keys at a time (whole 128-byte lines of V^T) but CONSUMED in 32-key halves u = 2 tile + sub: only
  // 32 x 64 scores and two 32-key P buffers are live, which is what lets Q, S, P and the fragments fit without spills
  // (a spilled value returns through scratch_load + vmcnt(0), i.e. it drains the DMA queue).
  // S^T of half-tile (tile, SUB): s[qh] = K[SUB] Q[qh]^T
  auto qk = [&](int tile, auto sub_c, f32x16 (&s)[2]) {   // prologue form (the loop inlines the same MFMAs step by step)
    constexpr int SUB = decltype(sub_c)::value;
    const char* Ks = k_ring + (tile & (NS - 1)) * K_TILE + k_row_off + SUB * 8192;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(Ks + (((2 * ks + h2) ^ k_sw) * 16));
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {   // Q from AGPRs, S into VGPRs: see the note in region()
        const bf16x8 qv = qf[qh][ks];
        if (ks == 0)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s[qh]) : "v"(kf), "a"(qv));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[qh]) : "v"(kf), "a"(qv));
      }
    }
    // hipcc does not see an MFMA inside the asm: cover the XDL-write -> VALU-read hazard (18 wait states) by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  };
  // O^T += V^T P^T over the 32 keys of half-tile (tile, SUB): kv blocks 2 SUB, 2 SUB + 1
  auto pv = [&](int tile, auto sub_c, const bf16x8 (&pf)[2][2]) {
    constexpr int SUB = decltype(sub_c)::value;
    const char* Vs = v_ring + (tile & (NS - 1)) * V_TILE + v_row_off;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = *(const bf16x8*)(Vs + dt * 4096 + (((2 * (2 * SUB + k2) + h2) ^ v_sw) * 16));
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) o_acc[qh][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qh][k2], o_acc[qh][dt], 0, 0, 0);
      }
  };
  typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
  // probabilities of one query half against the offset mc (= m_run * c), packed as the PV B operand; returns the row sum
  auto probs = [&](const f32x16& s, float mc, bf16x8 (&pf)[2]) -> float {
    float psum = 0.0f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p0 = __builtin_amdgcn_exp2f(s[8 * g + 2 * j] * c - mc);
        const float p1 = __builtin_amdgcn_exp2f(s[8 * g + 2 * j + 1] * c - mc);
        pk.u[j] = pack_bf2(p0, p1);
        psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]), __builtin_bit_cast(bf2v, 0x3f803f80u), psum,
                                              false);
      }
      pf[g] = pk.v;
    }
    return psum;
  };
  // exact path of the lazy running max (first half-tile, or a row sum outside [0, 2^40)): max, grow m, rescale, recompute
  auto fixup = [&](int qh, const f32x16& s, bf16x8 (&pf)[2], float& psum) {
    float mt = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float m_new = fmaxf(m_run[qh], mt);
    if (m_new == -INFINITY) {   // a fully masked half-tile in front of any real key cannot occur (masking is at the tail only)
      psum = 0.0f;
      return;
    }
    const float alpha = __builtin_amdgcn_exp2f((m_run[qh] - m_new) * c);
    m_run[qh] = m_new;
    l_run[qh] *= alpha;
    rescale_o(o_acc[qh], alpha, qh);
    psum = probs(s, m_new * c, pf);
  };
  auto mask_tail = [&](int kv_base, f32x16 (&s)[2]) {   // keys past Skv in the ragged last tile
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kv = kv_base + (e & 3) + 8 * (e >> 2) + 4 * h2;
        if (kv >= Skv) s[qh][e] = -INFINITY;
      }
  };
  auto finish_softmax = [&](f32x16 (&s)[2], bf16x8 (&pf)[2][2], float (&psum)[2]) {
    if (__any(!(psum[0] < 1.0995116e12f) || !(psum[1] < 1.0995116e12f))) {   // 2^40; also inf / NaN -- rare: one branch
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
        if (__any(!(psum[qh] < 1.0995116e12f))) fixup(qh, s[qh], pf[qh], psum[qh]);
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];
  };
  // tile boundary, at the top of the EVEN half-tile u = 2 t: K(t+1) and V(t) have landed (one DMA group stays in flight) --
  // a region late for its own reads (K(t) sub 1, V(t-1) sub 1), but the fragment prefetch at the end of this region already
  // reaches into K(t+1).  Every wave is past its reads of K(t-1) and V(t-2) (each was consumed by an MFMA behind a counted
  // wait), so after the barrier their slots take K(t+3) and V(t+2).
  auto boundary = [&](int t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);   // K(t + 3) and V(t + 2) go out piece by piece during the region (stage_piece)
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 se[2], so[2];      // S of the even / odd half-tile
  bf16x8 pe[2][2], po[2][2];
  // One pipelined half-tile u (CUR = u & 1): S of u + 1 and the PV of u - 1 under the softmax of u.  Written as 16 steps of
  // { fragment read for step + 3;  2 MFMAs;  one score pair of the softmax (2 fma, 2 exp2, pack, dot2) } with a scheduling
  // barrier after each step: hipcc's own ordering (and sched_group_barrier patterns) put every read right in front of its
  // MFMAs and the whole softmax behind them.
  // SL: the ring slot (tile & 3) of tile t = u >> 1 as a compile-time constant (the main loop is unrolled over four tiles), or
  // -1 for the runtime form (remainder tiles).  With SL known every fragment address is a per-lane constant plus an
  // IMMEDIATE offset (slot, half, d-tile): no vector add per read, and the DMA's LDS destination is an immediate M0.
  auto region = [&](int u, auto cur_c, auto sl_c, f32x16 (&sc)[2], f32x16 (&sn)[2], bf16x8 (&pc)[2][2],
                    const bf16x8 (&pp)[2][2]) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int SL = decltype(sl_c)::value;
    const int t = u >> 1;
    if (CUR == 0) boundary(t);
    if (ragged && t == n_tiles - 1) mask_tail(t * KVB + CUR * 32, sc);
    __builtin_amdgcn_sched_barrier(0);
    // u + 1 = (t, 1) and u - 1 = (t - 1, 1) for an even u;  (t + 1, 0) and (t, 0) for an odd one
    const int kt = CUR == 0 ? t : t + 1, vt_ = CUR == 0 ? t - 1 : t;
    constexpr int KSUB = CUR == 0 ? 1 : 0, VSUB = CUR == 0 ? 1 : 0;
    const uint32_t ks_base = lds0 + (kt & (NS - 1)) * K_TILE + k_row_off + KSUB * 8192;
    const uint32_t vs_base = lds0 + NS * K_TILE + (vt_ & (NS - 1)) * V_TILE + v_row_off;
    // the NEXT region's K fragments (its steps 0 - 2 are fetched by this region's steps 13 - 15): half-tile u + 2 = (t + 1, CUR)
    const uint32_t kn_base = lds0 + ((t + 1) & (NS - 1)) * K_TILE + k_row_off + CUR * 8192;
    // ... and its first V^T fragment (its step 1): the PV of half-tile u = (t, CUR)
    const uint32_t vn_base = lds0 + NS * K_TILE + (t & (NS - 1)) * V_TILE + v_row_off;
    constexpr int VNSUB = CUR;
    // Fragment reads and MFMAs are inline asm with hand-counted waits: hipcc answers every fragment dependence here with
    // s_waitcnt lgkmcnt(0) (a full LDS round trip every four steps), and moves O between the register files in front of the
    // (rare) rescale branch unless O is pinned to AccVGPRs.  LDS returns in order and nothing else uses the counter in the
    // loop: with three younger reads in flight, lgkmcnt(3) means "the fragment of this step has arrived".
    // Step order: QK and PV steps ALTERNATE (even ST: k-step ST / 2 of S_next, odd ST: PV block (ST - 1) / 2), so the two
    // S accumulators are touched every fourth MFMA instead of every second (no dependent-accumulate stall) .
    auto rd = [&](auto step_c) {   // the fragment of step ST (16 .. 18: steps 0 .. 2 of the next region) -> ring slot ST & 3
      constexpr int ST = decltype(step_c)::value;
      if constexpr (SL >= 0) {
        // slots: K of this region: tile t (CUR 0) or t + 1 (CUR 1); V: tile t - 1 or t; next region: K(t + 1), V(t)
        constexpr int KSL = CUR == 0 ? SL : (SL + 1) & 3, VSL = CUR == 0 ? (SL + 3) & 3 : SL;
        if constexpr (ST >= 16) {
          constexpr int S2 = ST - 16;
          if constexpr ((S2 & 1) == 0) {
            frag_read<ST & 3, ((SL + 1) & 3) * K_TILE + CUR * 8192>(kc[S2 >> 1]);
          } else {
            constexpr int k2 = (S2 >> 1) >> 2, dt = (S2 >> 1) & 3;
            frag_read<ST & 3, SL * V_TILE + dt * 4096>(vc[2 * VNSUB + k2]);
          }
        } else if constexpr ((ST & 1) == 0) {
          frag_read<ST & 3, KSL * K_TILE + KSUB * 8192>(kc[ST >> 1]);
        } else {
          constexpr int k2 = (ST >> 1) >> 2, dt = (ST >> 1) & 3;
          frag_read<ST & 3, VSL * V_TILE + dt * 4096>(vc[2 * VSUB + k2]);
        }
      } else if constexpr (ST >= 16) {
        constexpr int S2 = ST - 16;   // next region: step 0 = its K k-step 0, step 1 = ITS V block 0, step 2 = its K k-step 1
        if constexpr ((S2 & 1) == 0) {
          frag_read<ST & 3, 0>(kn_base + (((2 * (S2 >> 1) + h2) ^ k_sw) * 16));
        } else {
          constexpr int k2 = (S2 >> 1) >> 2, dt = (S2 >> 1) & 3;
          frag_read<ST & 3, dt * 4096>(vn_base + (((2 * (2 * VNSUB + k2) + h2) ^ v_sw) * 16));
        }
      } else if constexpr ((ST & 1) == 0) {
        frag_read<ST & 3, 0>(ks_base + (((2 * (ST >> 1) + h2) ^ k_sw) * 16));
      } else {
        constexpr int k2 = (ST >> 1) >> 2, dt = (ST >> 1) & 3;
        frag_read<ST & 3, dt * 4096>(vs_base + (((2 * (2 * VSUB + k2) + h2) ^ v_sw) * 16));
      }
    };
    const float mc[2] = {m_run[0] * c, m_run[1] * c};
    float psum[2] = {0.0f, 0.0f};
    union { bf16x8 v; uint32_t w[4]; } pk[2][2];
    auto step = [&](auto step_c) {
      constexpr int ST = decltype(step_c)::value;
      // ONE wave per SIMD issues in order: two MFMAs back to back stall the issue port for the 28 cycles the first one still
      // holds the matrix pipe, and the VALU work behind them then runs with the pipe idle.  So: MFMA, half of the score pair
      // (and the fragment read, whose address arithmetic is VALU too), MFMA, the other half -- each half fits the shadow.
      constexpr int qh = ST >> 3, g = (ST >> 2) & 1, jj = ST & 3;
#ifndef ALG_Q64_NO_READS
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // reads ST + 1, ST + 2 in flight: fragment ST has arrived
#endif
      constexpr bool QK = (ST & 1) == 0;
      constexpr int KS = ST >> 1;                       // k-step of S_next (QK steps)
      constexpr int k2 = (ST >> 1) >> 2, dt = (ST >> 1) & 3;   // kv block and d-tile (PV steps)
      if constexpr (QK) {
        // Register files, by hand: the Q fragments live in AccVGPRs and feed the MFMA from there, the scores land in
        // ArchVGPRs, where the VALU of the NEXT region reads them (a VALU operand cannot be an AccVGPR).  Left to itself hipcc
        // keeps Q in VGPRs, spills 37 of them to AGPRs and copies S out of AGPRs: 95 v_accvgpr_read per region next to 96
        // instructions of softmax.  The result is first read >= 16 MFMAs later (no XDL -> VALU hazard).
        qk_mfma<ST & 3, KS == 0>(sn[0], qf[0][KS]);
      } else {
        pv_mfma<dt, ST & 3>(o_acc[0][dt], pp[0][k2]);
      }
#ifdef ALG_Q64_NO_SOFTMAX
      const float a0 = 0.0f, a1 = 0.0f, p0 = 0.5f;
#elif defined(ALG_Q64_DUMMY_VALU)   // the same VALU instructions on a register no MFMA ever wrote
      float dm0 = mc[0], dm1 = mc[1];
      asm volatile("" : "+v"(dm0), "+v"(dm1));
      const float a0 = dm0 * c - mc[qh];
      const float a1 = dm1 * c - mc[qh];
      const float p0 = __builtin_amdgcn_exp2f(a0);
#else
      const float a0 = sc[qh][8 * g + 2 * jj] * c - mc[qh];
      const float a1 = sc[qh][8 * g + 2 * jj + 1] * c - mc[qh];
#ifdef ALG_Q64_NO_EXP
      const float p0 = a0;
#else
      const float p0 = __builtin_amdgcn_exp2f(a0);
#endif
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (QK) {
        qk_mfma<ST & 3, KS == 0>(sn[1], qf[1][KS]);
      } else {
        pv_mfma<4 + dt, ST & 3>(o_acc[1][dt], pp[1][k2]);
      }
      rd(std::integral_constant<int, ST + 3>{});   // into the slot of step ST - 1 (both of its MFMAs have been issued)
#ifndef ALG_Q64_NO_DMA
      if constexpr (CUR == 0 && (ST & 1) == 1) stage_piece(t + 3, t + 2, std::integral_constant<int, (ST >> 1)>{});
#endif
      {   // score pair ST of the softmax: query half ST >> 3, register quad g, pair jj
#ifdef ALG_Q64_NO_SOFTMAX
        pk[qh][g].w[jj] = 0x3f003f00u + (uint32_t)(a1 != 0.0f);
        psum[qh] = 1.0f;
#else
#ifdef ALG_Q64_NO_EXP
        const float p1 = a1;
#else
        const float p1 = __builtin_amdgcn_exp2f(a1);
#endif
#ifdef ALG_Q64_NO_PACK
        pk[qh][g].w[jj] = __float_as_uint(p0) ^ __float_as_uint(p1);
#else
        pk[qh][g].w[jj] = pack_bf2(p0, p1);
#endif
#ifndef ALG_Q64_NO_DOT2
        psum[qh] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk[qh][g].w[jj]),
                                                  __builtin_bit_cast(bf2v, 0x3f803f80u), psum[qh], false);
#else
        psum[qh] = 1.0f;
#endif
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
      for (int g = 0; g < 2; ++g) pc[qh][g] = pk[qh][g].v;
    finish_softmax(sc, pc, psum);
  };

  // ---- prologue: K(0), K(1), V(0), [K(2), V(1)], [K(3), V(2)] in the issue order the counted waits assume; S of half-tiles
  // 0 and 1 and the softmax of 0 un-pipelined; the first three fragments of region 1 ----
  stage_k(0);
  stage_k(1);
  stage_v(0);
  stage_k(2);
  stage_v(1);
  stage_k(3);
  stage_v(2);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // K(0)
  __builtin_amdgcn_s_barrier();
  qk(0, S0{}, se);
  {
    float psum[2];
    qk(0, S1{}, so);
    psum[0] = probs(se[0], m_run[0] * c, pe[0]);
    psum[1] = probs(se[1], m_run[1] * c, pe[1]);
    finish_softmax(se, pe, psum);
  }
  asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");   // K(1), V(0): what region 1 reads
  __builtin_amdgcn_s_barrier();
  {
    const uint32_t kb = lds0 + 1 * K_TILE + k_row_off;             // region 1, steps 0 - 2: K(1) sub 0, ks = 0, 1, 2
    const uint32_t vb = lds0 + NS * K_TILE + 0 * V_TILE + v_row_off;   // step 1: V(0) sub 0, kv block 0, d-tile 0
    frag_read<0, 0>(kb + (((0 + h2) ^ k_sw) * 16));
    frag_read<1, 0>(vb + (((0 + h2) ^ v_sw) * 16));
    frag_read<2, 0>(kb + (((2 + h2) ^ k_sw) * 16));
  }
  // u = 1, 2, ..., 2 n - 1; the S computed for u = 2 n (past the end) reads the re-fetched last tile and is dropped.
  // Pairs (2 t + 1, 2 t + 2) for t = 0 .. n - 2, four tiles per trip with their ring slots as constants, then the remainder
  // and the last odd half-tile in the runtime-slot form.
  using SLR = std::integral_constant<int, -1>;
  int t = 0;
  for (; t + 4 <= n_tiles - 1; t += 4) {   // t is a multiple of 4 here: tile t + i sits in slot i
    region(2 * t + 1, S1{}, std::integral_constant<int, 0>{}, so, se, po, pe);
    region(2 * t + 2, S0{}, std::integral_constant<int, 1>{}, se, so, pe, po);
    region(2 * t + 3, S1{}, std::integral_constant<int, 1>{}, so, se, po, pe);
    region(2 * t + 4, S0{}, std::integral_constant<int, 2>{}, se, so, pe, po);
    region(2 * t + 5, S1{}, std::integral_constant<int, 2>{}, so, se, po, pe);
    region(2 * t + 6, S0{}, std::integral_constant<int, 3>{}, se, so, pe, po);
    region(2 * t + 7, S1{}, std::integral_constant<int, 3>{}, so, se, po, pe);
    region(2 * t + 8, S0{}, std::integral_constant<int, 0>{}, se, so, pe, po);
  }
  for (; t < n_tiles - 1; ++t) {
    region(2 * t + 1, S1{}, SLR{}, so, se, po, pe);
    region(2 * t + 2, S0{}, SLR{}, se, so, pe, po);
  }
  region(2 * n_tiles - 1, S1{}, SLR{}, so, se, po, pe);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  pv(n_tiles - 1, S1{}, po);

#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const float l_tot = l_run[qh] + __shfl_xor(l_run[qh], 32, 64);
    const float inv = 1.0f / l_tot;
    const int q_row = q_row0 + qh * 32;
    if (q_row < Sq) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 128;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * h2;
          uint2 v;
          v.x = pack_bf2(o_acc[qh][dt][4 * g] * inv, o_acc[qh][dt][4 * g + 1] * inv);
          v.y = pack_bf2(o_acc[qh][dt][4 * g + 2] * inv, o_acc[qh][dt][4 * g + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
}

}  // namespace a128q

// Returns ALG_OK when launched, 1 when this call is not covered (the caller runs attention128.hip's kernel).
int flash_attn_d128_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                        int64_t o_rs, float scale, hipStream_t stream) {
  using namespace a128q;
  const char* env = getenv("ALG_ATTN128_Q64");   // 0: keep attention128.hip's 32-query kernel (A/B runs, bit-level comparisons)
  const int enabled = env ? atoi(env) : 1;
  if (!enabled || (Skv + KVB - 1) / KVB < MIN_TILES) return 1;
  // 31-bit BYTE offsets inside one (batch, head) for the buffer-load DMA; V^T rows cover whole 64-key tiles
  if ((int64_t)(Skv + 64) * k_rs * 2 >= (1ll << 31) || (int64_t)129 * vt_rs * 2 >= (1ll << 31)) return 1;
  if (vt_rs < (int64_t)((Skv + KVB - 1) / KVB) * KVB) return 1;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)flash_attn_d128_q64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess)
      return 1;
    attr_set = true;
  }
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + NW * QW - 1) / (NW * QW);
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.scale_log2 = scale * 1.4426950408889634f;
  const int64_t grid = (int64_t)((batch * heads + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) return 1;
  hipLaunchKernelGGL(flash_attn_d128_q64_kernel, dim3((unsigned)grid), dim3(NW * 64), LDS_BYTES, stream, p);
  return check_launch("alg_flash_attn_d128");
}

}  // namespace alg
