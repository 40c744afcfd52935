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
// DEFAULT since round 4 for non-causal, ungrouped attention over at least POLICY_TILES KV tiles (the Wan / HunyuanVideo
// self-attention: +3.2 % over attention128_pipe.hip at C4); ALG_ATTN128_Q64=0 switches it off, =2 takes every call of at
// least MIN_TILES tiles (tests).  The cross-attentions, the causal grouped-query form and short sequences stay on
// attention128_pipe.hip / attention128.hip.  Rounds 2 - 4 kept it out of the product because 1 - 2 % of fp8 C5 forwards differed
// from their repeat; the cause (an asm MFMA's dead destination registers recycled by hipcc while the MFMA still writes them) is
// described where it is fixed, at the end of the kernel, and in profiles/r4_attention128_q64_probe.txt.  Same operand layout, same
// swizzles, same accumulation order per query as that kernel (S^T = K Q^T, P as the B operand of O^T = V^T P^T).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// Row-sum limit of the lazy running max: 2^40 (what the kernel was validated with).  Unlike the single-statement kernels, whose
// waves leave the fast path for good (hence their 2^80), the exact path here is an in-line branch that returns to the fast path:
// no cliff, so the smaller limit costs nothing on model data.
#define ALG_Q64_SUM_LIMIT 1.0995116e12f

namespace alg {
namespace a128q {

constexpr int NW = 4;
constexpr int QW = 64;                   // queries per wave
constexpr int KVB = 64;
constexpr int K_TILE = KVB * 128 * 2;    // 16 KiB
constexpr int V_TILE = 128 * KVB * 2;    // 16 KiB
constexpr int NS = 4;                    // ring slots per operand
constexpr int LDS_BYTES = NS * (K_TILE + V_TILE);
constexpr int MIN_TILES = 8;             // what the kernel can take (ALG_ATTN128_Q64=2)
constexpr int POLICY_TILES = 64;         // what it takes by default: 4,096 keys and more
}  // namespace a128q
extern std::atomic<uint64_t*> g_clock_tap;
extern std::atomic<int> g_clock_tap_slots;
namespace a128q {

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
#ifdef ALG_EXPERIMENTS
  float* dbg;   // investigation tap (alg_debug_q64_tap): per (batch, head, query) [l_run, m_run, l_tot, 1 / l_tot] per lane
#endif
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
// O is NOT a C++ value inside the loop: it lives in a[0:127] (tile (qh, dt) = a[16 (4 qh + dt) .. + 15]) and is named literally by
// every asm that touches it (all of them list a0 - a127 as clobbers).  As an asm OPERAND pinned to those registers it is at
// the allocator's mercy: in the d = 64 sibling hipcc kept such an O in ArchVGPRs between the asms -- 16 v_accvgpr_write in
// front of every MFMA and reads right behind it, i.e. behind an MFMA it cannot see (wrong results, not just slow).
#define ALG_O_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
template <int IDX, int SLOT>
__device__ __forceinline__ void pv_mfma(const bf16x8 pfrag) {
#define ALG_PV2(LO, HI, FR) asm volatile("v_mfma_f32_32x32x16_bf16 a[" #LO ":" #HI "], " FR ", %0, a[" #LO ":" #HI "]" ::"v"(pfrag) : ALG_O_CLOBBER, ALG_FRAG_CLOBBER);
#define ALG_PV_CASE(I, LO, HI)                        \
  if constexpr (IDX == I) {                           \
    if constexpr (SLOT == 0) { ALG_PV2(LO, HI, ALG_FR0) } \
    if constexpr (SLOT == 1) { ALG_PV2(LO, HI, ALG_FR1) } \
    if constexpr (SLOT == 2) { ALG_PV2(LO, HI, ALG_FR2) } \
    if constexpr (SLOT == 3) { ALG_PV2(LO, HI, ALG_FR3) } \
  }
  ALG_PV_CASE(0, 0, 15) ALG_PV_CASE(1, 16, 31) ALG_PV_CASE(2, 32, 47) ALG_PV_CASE(3, 48, 63) ALG_PV_CASE(4, 64, 79) ALG_PV_CASE(5, 80, 95) ALG_PV_CASE(6, 96, 111) ALG_PV_CASE(7, 112, 127)
#undef ALG_PV_CASE
#undef ALG_PV2
}
__device__ __forceinline__ void zero_o() {
  asm volatile("v_accvgpr_write_b32 a0, 0\n\t"
               "v_accvgpr_write_b32 a1, 0\n\t"
               "v_accvgpr_write_b32 a2, 0\n\t"
               "v_accvgpr_write_b32 a3, 0\n\t"
               "v_accvgpr_write_b32 a4, 0\n\t"
               "v_accvgpr_write_b32 a5, 0\n\t"
               "v_accvgpr_write_b32 a6, 0\n\t"
               "v_accvgpr_write_b32 a7, 0\n\t"
               "v_accvgpr_write_b32 a8, 0\n\t"
               "v_accvgpr_write_b32 a9, 0\n\t"
               "v_accvgpr_write_b32 a10, 0\n\t"
               "v_accvgpr_write_b32 a11, 0\n\t"
               "v_accvgpr_write_b32 a12, 0\n\t"
               "v_accvgpr_write_b32 a13, 0\n\t"
               "v_accvgpr_write_b32 a14, 0\n\t"
               "v_accvgpr_write_b32 a15, 0\n\t"
               "v_accvgpr_write_b32 a16, 0\n\t"
               "v_accvgpr_write_b32 a17, 0\n\t"
               "v_accvgpr_write_b32 a18, 0\n\t"
               "v_accvgpr_write_b32 a19, 0\n\t"
               "v_accvgpr_write_b32 a20, 0\n\t"
               "v_accvgpr_write_b32 a21, 0\n\t"
               "v_accvgpr_write_b32 a22, 0\n\t"
               "v_accvgpr_write_b32 a23, 0\n\t"
               "v_accvgpr_write_b32 a24, 0\n\t"
               "v_accvgpr_write_b32 a25, 0\n\t"
               "v_accvgpr_write_b32 a26, 0\n\t"
               "v_accvgpr_write_b32 a27, 0\n\t"
               "v_accvgpr_write_b32 a28, 0\n\t"
               "v_accvgpr_write_b32 a29, 0\n\t"
               "v_accvgpr_write_b32 a30, 0\n\t"
               "v_accvgpr_write_b32 a31, 0\n\t"
               "v_accvgpr_write_b32 a32, 0\n\t"
               "v_accvgpr_write_b32 a33, 0\n\t"
               "v_accvgpr_write_b32 a34, 0\n\t"
               "v_accvgpr_write_b32 a35, 0\n\t"
               "v_accvgpr_write_b32 a36, 0\n\t"
               "v_accvgpr_write_b32 a37, 0\n\t"
               "v_accvgpr_write_b32 a38, 0\n\t"
               "v_accvgpr_write_b32 a39, 0\n\t"
               "v_accvgpr_write_b32 a40, 0\n\t"
               "v_accvgpr_write_b32 a41, 0\n\t"
               "v_accvgpr_write_b32 a42, 0\n\t"
               "v_accvgpr_write_b32 a43, 0\n\t"
               "v_accvgpr_write_b32 a44, 0\n\t"
               "v_accvgpr_write_b32 a45, 0\n\t"
               "v_accvgpr_write_b32 a46, 0\n\t"
               "v_accvgpr_write_b32 a47, 0\n\t"
               "v_accvgpr_write_b32 a48, 0\n\t"
               "v_accvgpr_write_b32 a49, 0\n\t"
               "v_accvgpr_write_b32 a50, 0\n\t"
               "v_accvgpr_write_b32 a51, 0\n\t"
               "v_accvgpr_write_b32 a52, 0\n\t"
               "v_accvgpr_write_b32 a53, 0\n\t"
               "v_accvgpr_write_b32 a54, 0\n\t"
               "v_accvgpr_write_b32 a55, 0\n\t"
               "v_accvgpr_write_b32 a56, 0\n\t"
               "v_accvgpr_write_b32 a57, 0\n\t"
               "v_accvgpr_write_b32 a58, 0\n\t"
               "v_accvgpr_write_b32 a59, 0\n\t"
               "v_accvgpr_write_b32 a60, 0\n\t"
               "v_accvgpr_write_b32 a61, 0\n\t"
               "v_accvgpr_write_b32 a62, 0\n\t"
               "v_accvgpr_write_b32 a63, 0\n\t"
               "v_accvgpr_write_b32 a64, 0\n\t"
               "v_accvgpr_write_b32 a65, 0\n\t"
               "v_accvgpr_write_b32 a66, 0\n\t"
               "v_accvgpr_write_b32 a67, 0\n\t"
               "v_accvgpr_write_b32 a68, 0\n\t"
               "v_accvgpr_write_b32 a69, 0\n\t"
               "v_accvgpr_write_b32 a70, 0\n\t"
               "v_accvgpr_write_b32 a71, 0\n\t"
               "v_accvgpr_write_b32 a72, 0\n\t"
               "v_accvgpr_write_b32 a73, 0\n\t"
               "v_accvgpr_write_b32 a74, 0\n\t"
               "v_accvgpr_write_b32 a75, 0\n\t"
               "v_accvgpr_write_b32 a76, 0\n\t"
               "v_accvgpr_write_b32 a77, 0\n\t"
               "v_accvgpr_write_b32 a78, 0\n\t"
               "v_accvgpr_write_b32 a79, 0\n\t"
               "v_accvgpr_write_b32 a80, 0\n\t"
               "v_accvgpr_write_b32 a81, 0\n\t"
               "v_accvgpr_write_b32 a82, 0\n\t"
               "v_accvgpr_write_b32 a83, 0\n\t"
               "v_accvgpr_write_b32 a84, 0\n\t"
               "v_accvgpr_write_b32 a85, 0\n\t"
               "v_accvgpr_write_b32 a86, 0\n\t"
               "v_accvgpr_write_b32 a87, 0\n\t"
               "v_accvgpr_write_b32 a88, 0\n\t"
               "v_accvgpr_write_b32 a89, 0\n\t"
               "v_accvgpr_write_b32 a90, 0\n\t"
               "v_accvgpr_write_b32 a91, 0\n\t"
               "v_accvgpr_write_b32 a92, 0\n\t"
               "v_accvgpr_write_b32 a93, 0\n\t"
               "v_accvgpr_write_b32 a94, 0\n\t"
               "v_accvgpr_write_b32 a95, 0\n\t"
               "v_accvgpr_write_b32 a96, 0\n\t"
               "v_accvgpr_write_b32 a97, 0\n\t"
               "v_accvgpr_write_b32 a98, 0\n\t"
               "v_accvgpr_write_b32 a99, 0\n\t"
               "v_accvgpr_write_b32 a100, 0\n\t"
               "v_accvgpr_write_b32 a101, 0\n\t"
               "v_accvgpr_write_b32 a102, 0\n\t"
               "v_accvgpr_write_b32 a103, 0\n\t"
               "v_accvgpr_write_b32 a104, 0\n\t"
               "v_accvgpr_write_b32 a105, 0\n\t"
               "v_accvgpr_write_b32 a106, 0\n\t"
               "v_accvgpr_write_b32 a107, 0\n\t"
               "v_accvgpr_write_b32 a108, 0\n\t"
               "v_accvgpr_write_b32 a109, 0\n\t"
               "v_accvgpr_write_b32 a110, 0\n\t"
               "v_accvgpr_write_b32 a111, 0\n\t"
               "v_accvgpr_write_b32 a112, 0\n\t"
               "v_accvgpr_write_b32 a113, 0\n\t"
               "v_accvgpr_write_b32 a114, 0\n\t"
               "v_accvgpr_write_b32 a115, 0\n\t"
               "v_accvgpr_write_b32 a116, 0\n\t"
               "v_accvgpr_write_b32 a117, 0\n\t"
               "v_accvgpr_write_b32 a118, 0\n\t"
               "v_accvgpr_write_b32 a119, 0\n\t"
               "v_accvgpr_write_b32 a120, 0\n\t"
               "v_accvgpr_write_b32 a121, 0\n\t"
               "v_accvgpr_write_b32 a122, 0\n\t"
               "v_accvgpr_write_b32 a123, 0\n\t"
               "v_accvgpr_write_b32 a124, 0\n\t"
               "v_accvgpr_write_b32 a125, 0\n\t"
               "v_accvgpr_write_b32 a126, 0\n\t"
               "v_accvgpr_write_b32 a127, 0\n\t"
               "s_nop 0" ::: ALG_O_CLOBBER);
}
// O[qh] *= alpha (leading s_nops: XDL write -> accvgpr_read hazard, invisible to hipcc)
__device__ __forceinline__ void rescale_o(float alpha, int qh) {
  float tmp;
  if (qh == 0) {
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
                 "v_accvgpr_read_b32 %0, a0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a0, %0\n\t"
                 "v_accvgpr_read_b32 %0, a1\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a1, %0\n\t"
                 "v_accvgpr_read_b32 %0, a2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a2, %0\n\t"
                 "v_accvgpr_read_b32 %0, a3\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a3, %0\n\t"
                 "v_accvgpr_read_b32 %0, a4\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a4, %0\n\t"
                 "v_accvgpr_read_b32 %0, a5\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a5, %0\n\t"
                 "v_accvgpr_read_b32 %0, a6\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a6, %0\n\t"
                 "v_accvgpr_read_b32 %0, a7\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a7, %0\n\t"
                 "v_accvgpr_read_b32 %0, a8\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a8, %0\n\t"
                 "v_accvgpr_read_b32 %0, a9\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a9, %0\n\t"
                 "v_accvgpr_read_b32 %0, a10\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a10, %0\n\t"
                 "v_accvgpr_read_b32 %0, a11\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a11, %0\n\t"
                 "v_accvgpr_read_b32 %0, a12\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a12, %0\n\t"
                 "v_accvgpr_read_b32 %0, a13\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a13, %0\n\t"
                 "v_accvgpr_read_b32 %0, a14\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a14, %0\n\t"
                 "v_accvgpr_read_b32 %0, a15\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a15, %0\n\t"
                 "v_accvgpr_read_b32 %0, a16\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a16, %0\n\t"
                 "v_accvgpr_read_b32 %0, a17\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a17, %0\n\t"
                 "v_accvgpr_read_b32 %0, a18\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a18, %0\n\t"
                 "v_accvgpr_read_b32 %0, a19\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a19, %0\n\t"
                 "v_accvgpr_read_b32 %0, a20\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a20, %0\n\t"
                 "v_accvgpr_read_b32 %0, a21\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a21, %0\n\t"
                 "v_accvgpr_read_b32 %0, a22\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a22, %0\n\t"
                 "v_accvgpr_read_b32 %0, a23\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a23, %0\n\t"
                 "v_accvgpr_read_b32 %0, a24\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a24, %0\n\t"
                 "v_accvgpr_read_b32 %0, a25\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a25, %0\n\t"
                 "v_accvgpr_read_b32 %0, a26\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a26, %0\n\t"
                 "v_accvgpr_read_b32 %0, a27\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a27, %0\n\t"
                 "v_accvgpr_read_b32 %0, a28\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a28, %0\n\t"
                 "v_accvgpr_read_b32 %0, a29\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a29, %0\n\t"
                 "v_accvgpr_read_b32 %0, a30\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a30, %0\n\t"
                 "v_accvgpr_read_b32 %0, a31\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a31, %0\n\t"
                 "v_accvgpr_read_b32 %0, a32\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a32, %0\n\t"
                 "v_accvgpr_read_b32 %0, a33\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a33, %0\n\t"
                 "v_accvgpr_read_b32 %0, a34\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a34, %0\n\t"
                 "v_accvgpr_read_b32 %0, a35\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a35, %0\n\t"
                 "v_accvgpr_read_b32 %0, a36\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a36, %0\n\t"
                 "v_accvgpr_read_b32 %0, a37\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a37, %0\n\t"
                 "v_accvgpr_read_b32 %0, a38\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a38, %0\n\t"
                 "v_accvgpr_read_b32 %0, a39\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a39, %0\n\t"
                 "v_accvgpr_read_b32 %0, a40\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a40, %0\n\t"
                 "v_accvgpr_read_b32 %0, a41\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a41, %0\n\t"
                 "v_accvgpr_read_b32 %0, a42\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a42, %0\n\t"
                 "v_accvgpr_read_b32 %0, a43\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a43, %0\n\t"
                 "v_accvgpr_read_b32 %0, a44\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a44, %0\n\t"
                 "v_accvgpr_read_b32 %0, a45\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a45, %0\n\t"
                 "v_accvgpr_read_b32 %0, a46\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a46, %0\n\t"
                 "v_accvgpr_read_b32 %0, a47\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a47, %0\n\t"
                 "v_accvgpr_read_b32 %0, a48\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a48, %0\n\t"
                 "v_accvgpr_read_b32 %0, a49\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a49, %0\n\t"
                 "v_accvgpr_read_b32 %0, a50\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a50, %0\n\t"
                 "v_accvgpr_read_b32 %0, a51\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a51, %0\n\t"
                 "v_accvgpr_read_b32 %0, a52\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a52, %0\n\t"
                 "v_accvgpr_read_b32 %0, a53\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a53, %0\n\t"
                 "v_accvgpr_read_b32 %0, a54\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a54, %0\n\t"
                 "v_accvgpr_read_b32 %0, a55\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a55, %0\n\t"
                 "v_accvgpr_read_b32 %0, a56\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a56, %0\n\t"
                 "v_accvgpr_read_b32 %0, a57\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a57, %0\n\t"
                 "v_accvgpr_read_b32 %0, a58\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a58, %0\n\t"
                 "v_accvgpr_read_b32 %0, a59\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a59, %0\n\t"
                 "v_accvgpr_read_b32 %0, a60\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a60, %0\n\t"
                 "v_accvgpr_read_b32 %0, a61\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a61, %0\n\t"
                 "v_accvgpr_read_b32 %0, a62\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a62, %0\n\t"
                 "v_accvgpr_read_b32 %0, a63\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a63, %0\n\t"
                 "s_nop 0"
                 : "=&v"(tmp) : "v"(alpha) : ALG_O_CLOBBER);
  } else {
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
                 "v_accvgpr_read_b32 %0, a64\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a64, %0\n\t"
                 "v_accvgpr_read_b32 %0, a65\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a65, %0\n\t"
                 "v_accvgpr_read_b32 %0, a66\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a66, %0\n\t"
                 "v_accvgpr_read_b32 %0, a67\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a67, %0\n\t"
                 "v_accvgpr_read_b32 %0, a68\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a68, %0\n\t"
                 "v_accvgpr_read_b32 %0, a69\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a69, %0\n\t"
                 "v_accvgpr_read_b32 %0, a70\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a70, %0\n\t"
                 "v_accvgpr_read_b32 %0, a71\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a71, %0\n\t"
                 "v_accvgpr_read_b32 %0, a72\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a72, %0\n\t"
                 "v_accvgpr_read_b32 %0, a73\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a73, %0\n\t"
                 "v_accvgpr_read_b32 %0, a74\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a74, %0\n\t"
                 "v_accvgpr_read_b32 %0, a75\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a75, %0\n\t"
                 "v_accvgpr_read_b32 %0, a76\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a76, %0\n\t"
                 "v_accvgpr_read_b32 %0, a77\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a77, %0\n\t"
                 "v_accvgpr_read_b32 %0, a78\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a78, %0\n\t"
                 "v_accvgpr_read_b32 %0, a79\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a79, %0\n\t"
                 "v_accvgpr_read_b32 %0, a80\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a80, %0\n\t"
                 "v_accvgpr_read_b32 %0, a81\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a81, %0\n\t"
                 "v_accvgpr_read_b32 %0, a82\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a82, %0\n\t"
                 "v_accvgpr_read_b32 %0, a83\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a83, %0\n\t"
                 "v_accvgpr_read_b32 %0, a84\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a84, %0\n\t"
                 "v_accvgpr_read_b32 %0, a85\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a85, %0\n\t"
                 "v_accvgpr_read_b32 %0, a86\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a86, %0\n\t"
                 "v_accvgpr_read_b32 %0, a87\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a87, %0\n\t"
                 "v_accvgpr_read_b32 %0, a88\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a88, %0\n\t"
                 "v_accvgpr_read_b32 %0, a89\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a89, %0\n\t"
                 "v_accvgpr_read_b32 %0, a90\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a90, %0\n\t"
                 "v_accvgpr_read_b32 %0, a91\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a91, %0\n\t"
                 "v_accvgpr_read_b32 %0, a92\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a92, %0\n\t"
                 "v_accvgpr_read_b32 %0, a93\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a93, %0\n\t"
                 "v_accvgpr_read_b32 %0, a94\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a94, %0\n\t"
                 "v_accvgpr_read_b32 %0, a95\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a95, %0\n\t"
                 "v_accvgpr_read_b32 %0, a96\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a96, %0\n\t"
                 "v_accvgpr_read_b32 %0, a97\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a97, %0\n\t"
                 "v_accvgpr_read_b32 %0, a98\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a98, %0\n\t"
                 "v_accvgpr_read_b32 %0, a99\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a99, %0\n\t"
                 "v_accvgpr_read_b32 %0, a100\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a100, %0\n\t"
                 "v_accvgpr_read_b32 %0, a101\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a101, %0\n\t"
                 "v_accvgpr_read_b32 %0, a102\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a102, %0\n\t"
                 "v_accvgpr_read_b32 %0, a103\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a103, %0\n\t"
                 "v_accvgpr_read_b32 %0, a104\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a104, %0\n\t"
                 "v_accvgpr_read_b32 %0, a105\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a105, %0\n\t"
                 "v_accvgpr_read_b32 %0, a106\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a106, %0\n\t"
                 "v_accvgpr_read_b32 %0, a107\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a107, %0\n\t"
                 "v_accvgpr_read_b32 %0, a108\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a108, %0\n\t"
                 "v_accvgpr_read_b32 %0, a109\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a109, %0\n\t"
                 "v_accvgpr_read_b32 %0, a110\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a110, %0\n\t"
                 "v_accvgpr_read_b32 %0, a111\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a111, %0\n\t"
                 "v_accvgpr_read_b32 %0, a112\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a112, %0\n\t"
                 "v_accvgpr_read_b32 %0, a113\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a113, %0\n\t"
                 "v_accvgpr_read_b32 %0, a114\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a114, %0\n\t"
                 "v_accvgpr_read_b32 %0, a115\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a115, %0\n\t"
                 "v_accvgpr_read_b32 %0, a116\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a116, %0\n\t"
                 "v_accvgpr_read_b32 %0, a117\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a117, %0\n\t"
                 "v_accvgpr_read_b32 %0, a118\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a118, %0\n\t"
                 "v_accvgpr_read_b32 %0, a119\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a119, %0\n\t"
                 "v_accvgpr_read_b32 %0, a120\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a120, %0\n\t"
                 "v_accvgpr_read_b32 %0, a121\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a121, %0\n\t"
                 "v_accvgpr_read_b32 %0, a122\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a122, %0\n\t"
                 "v_accvgpr_read_b32 %0, a123\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a123, %0\n\t"
                 "v_accvgpr_read_b32 %0, a124\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a124, %0\n\t"
                 "v_accvgpr_read_b32 %0, a125\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a125, %0\n\t"
                 "v_accvgpr_read_b32 %0, a126\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a126, %0\n\t"
                 "v_accvgpr_read_b32 %0, a127\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a127, %0\n\t"
                 "s_nop 0"
                 : "=&v"(tmp) : "v"(alpha) : ALG_O_CLOBBER);
  }
}
// read one O tile out (after the loop; the caller has waited out the last MFMAs)
template <int IDX>
__device__ __forceinline__ void read_o(float (&f)[16]) {
  if constexpr (IDX == 0)
    asm volatile("v_accvgpr_read_b32 %0, a0\n\t"
                 "v_accvgpr_read_b32 %1, a1\n\t"
                 "v_accvgpr_read_b32 %2, a2\n\t"
                 "v_accvgpr_read_b32 %3, a3\n\t"
                 "v_accvgpr_read_b32 %4, a4\n\t"
                 "v_accvgpr_read_b32 %5, a5\n\t"
                 "v_accvgpr_read_b32 %6, a6\n\t"
                 "v_accvgpr_read_b32 %7, a7\n\t"
                 "v_accvgpr_read_b32 %8, a8\n\t"
                 "v_accvgpr_read_b32 %9, a9\n\t"
                 "v_accvgpr_read_b32 %10, a10\n\t"
                 "v_accvgpr_read_b32 %11, a11\n\t"
                 "v_accvgpr_read_b32 %12, a12\n\t"
                 "v_accvgpr_read_b32 %13, a13\n\t"
                 "v_accvgpr_read_b32 %14, a14\n\t"
                 "v_accvgpr_read_b32 %15, a15\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 1)
    asm volatile("v_accvgpr_read_b32 %0, a16\n\t"
                 "v_accvgpr_read_b32 %1, a17\n\t"
                 "v_accvgpr_read_b32 %2, a18\n\t"
                 "v_accvgpr_read_b32 %3, a19\n\t"
                 "v_accvgpr_read_b32 %4, a20\n\t"
                 "v_accvgpr_read_b32 %5, a21\n\t"
                 "v_accvgpr_read_b32 %6, a22\n\t"
                 "v_accvgpr_read_b32 %7, a23\n\t"
                 "v_accvgpr_read_b32 %8, a24\n\t"
                 "v_accvgpr_read_b32 %9, a25\n\t"
                 "v_accvgpr_read_b32 %10, a26\n\t"
                 "v_accvgpr_read_b32 %11, a27\n\t"
                 "v_accvgpr_read_b32 %12, a28\n\t"
                 "v_accvgpr_read_b32 %13, a29\n\t"
                 "v_accvgpr_read_b32 %14, a30\n\t"
                 "v_accvgpr_read_b32 %15, a31\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 2)
    asm volatile("v_accvgpr_read_b32 %0, a32\n\t"
                 "v_accvgpr_read_b32 %1, a33\n\t"
                 "v_accvgpr_read_b32 %2, a34\n\t"
                 "v_accvgpr_read_b32 %3, a35\n\t"
                 "v_accvgpr_read_b32 %4, a36\n\t"
                 "v_accvgpr_read_b32 %5, a37\n\t"
                 "v_accvgpr_read_b32 %6, a38\n\t"
                 "v_accvgpr_read_b32 %7, a39\n\t"
                 "v_accvgpr_read_b32 %8, a40\n\t"
                 "v_accvgpr_read_b32 %9, a41\n\t"
                 "v_accvgpr_read_b32 %10, a42\n\t"
                 "v_accvgpr_read_b32 %11, a43\n\t"
                 "v_accvgpr_read_b32 %12, a44\n\t"
                 "v_accvgpr_read_b32 %13, a45\n\t"
                 "v_accvgpr_read_b32 %14, a46\n\t"
                 "v_accvgpr_read_b32 %15, a47\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 3)
    asm volatile("v_accvgpr_read_b32 %0, a48\n\t"
                 "v_accvgpr_read_b32 %1, a49\n\t"
                 "v_accvgpr_read_b32 %2, a50\n\t"
                 "v_accvgpr_read_b32 %3, a51\n\t"
                 "v_accvgpr_read_b32 %4, a52\n\t"
                 "v_accvgpr_read_b32 %5, a53\n\t"
                 "v_accvgpr_read_b32 %6, a54\n\t"
                 "v_accvgpr_read_b32 %7, a55\n\t"
                 "v_accvgpr_read_b32 %8, a56\n\t"
                 "v_accvgpr_read_b32 %9, a57\n\t"
                 "v_accvgpr_read_b32 %10, a58\n\t"
                 "v_accvgpr_read_b32 %11, a59\n\t"
                 "v_accvgpr_read_b32 %12, a60\n\t"
                 "v_accvgpr_read_b32 %13, a61\n\t"
                 "v_accvgpr_read_b32 %14, a62\n\t"
                 "v_accvgpr_read_b32 %15, a63\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 4)
    asm volatile("v_accvgpr_read_b32 %0, a64\n\t"
                 "v_accvgpr_read_b32 %1, a65\n\t"
                 "v_accvgpr_read_b32 %2, a66\n\t"
                 "v_accvgpr_read_b32 %3, a67\n\t"
                 "v_accvgpr_read_b32 %4, a68\n\t"
                 "v_accvgpr_read_b32 %5, a69\n\t"
                 "v_accvgpr_read_b32 %6, a70\n\t"
                 "v_accvgpr_read_b32 %7, a71\n\t"
                 "v_accvgpr_read_b32 %8, a72\n\t"
                 "v_accvgpr_read_b32 %9, a73\n\t"
                 "v_accvgpr_read_b32 %10, a74\n\t"
                 "v_accvgpr_read_b32 %11, a75\n\t"
                 "v_accvgpr_read_b32 %12, a76\n\t"
                 "v_accvgpr_read_b32 %13, a77\n\t"
                 "v_accvgpr_read_b32 %14, a78\n\t"
                 "v_accvgpr_read_b32 %15, a79\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 5)
    asm volatile("v_accvgpr_read_b32 %0, a80\n\t"
                 "v_accvgpr_read_b32 %1, a81\n\t"
                 "v_accvgpr_read_b32 %2, a82\n\t"
                 "v_accvgpr_read_b32 %3, a83\n\t"
                 "v_accvgpr_read_b32 %4, a84\n\t"
                 "v_accvgpr_read_b32 %5, a85\n\t"
                 "v_accvgpr_read_b32 %6, a86\n\t"
                 "v_accvgpr_read_b32 %7, a87\n\t"
                 "v_accvgpr_read_b32 %8, a88\n\t"
                 "v_accvgpr_read_b32 %9, a89\n\t"
                 "v_accvgpr_read_b32 %10, a90\n\t"
                 "v_accvgpr_read_b32 %11, a91\n\t"
                 "v_accvgpr_read_b32 %12, a92\n\t"
                 "v_accvgpr_read_b32 %13, a93\n\t"
                 "v_accvgpr_read_b32 %14, a94\n\t"
                 "v_accvgpr_read_b32 %15, a95\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 6)
    asm volatile("v_accvgpr_read_b32 %0, a96\n\t"
                 "v_accvgpr_read_b32 %1, a97\n\t"
                 "v_accvgpr_read_b32 %2, a98\n\t"
                 "v_accvgpr_read_b32 %3, a99\n\t"
                 "v_accvgpr_read_b32 %4, a100\n\t"
                 "v_accvgpr_read_b32 %5, a101\n\t"
                 "v_accvgpr_read_b32 %6, a102\n\t"
                 "v_accvgpr_read_b32 %7, a103\n\t"
                 "v_accvgpr_read_b32 %8, a104\n\t"
                 "v_accvgpr_read_b32 %9, a105\n\t"
                 "v_accvgpr_read_b32 %10, a106\n\t"
                 "v_accvgpr_read_b32 %11, a107\n\t"
                 "v_accvgpr_read_b32 %12, a108\n\t"
                 "v_accvgpr_read_b32 %13, a109\n\t"
                 "v_accvgpr_read_b32 %14, a110\n\t"
                 "v_accvgpr_read_b32 %15, a111\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
  if constexpr (IDX == 7)
    asm volatile("v_accvgpr_read_b32 %0, a112\n\t"
                 "v_accvgpr_read_b32 %1, a113\n\t"
                 "v_accvgpr_read_b32 %2, a114\n\t"
                 "v_accvgpr_read_b32 %3, a115\n\t"
                 "v_accvgpr_read_b32 %4, a116\n\t"
                 "v_accvgpr_read_b32 %5, a117\n\t"
                 "v_accvgpr_read_b32 %6, a118\n\t"
                 "v_accvgpr_read_b32 %7, a119\n\t"
                 "v_accvgpr_read_b32 %8, a120\n\t"
                 "v_accvgpr_read_b32 %9, a121\n\t"
                 "v_accvgpr_read_b32 %10, a122\n\t"
                 "v_accvgpr_read_b32 %11, a123\n\t"
                 "v_accvgpr_read_b32 %12, a124\n\t"
                 "v_accvgpr_read_b32 %13, a125\n\t"
                 "v_accvgpr_read_b32 %14, a126\n\t"
                 "v_accvgpr_read_b32 %15, a127\n\t"
                 "s_nop 0"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ALG_O_CLOBBER);
}

// MODE 1 is the kernel.  2 - 4 are round 3's experiment arms (EXPERIMENTS build, ALG_ATTN128_Q64 = 12 / 13 / 14):
//   1  the round-2 kernel: counted vmcnt(8) at the tile boundary (three DMA groups in the ring, one may still be in flight)
//   2  vmcnt(0) at the tile boundary: no reliance on LDS-DMA completing in issue order (the group issued one tile ago is
//      ~2,000 cycles old by then)
//   3  MODE 1 + every wave drains its output stores and writes the L2 back (agent-scope release) before it ends
//   4  both
template <int MODE>
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
  const bool tap = p.clk != nullptr && (blockIdx.x & 63) == 0 && wave == 0;   // clock tap: see attention.hip
  uint64_t tap_c0 = 0, tap_r0 = 0;
  if (tap) {
    tap_c0 = __builtin_readcyclecounter();
    tap_r0 = wall_clock64();
  }
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

  // K tile: 64 rows x 16 slots (256 B rows), four rounds of 16 rows; physical slot tid & 15 holds logical slot
  // (tid & 15) ^ (row & 15).  V^T tile: 128 d-rows x 8 slots, four rounds of 32 rows, swizzle (row >> 1) & 7.
  // Tiles past the end re-fetch the last one (uniform instruction counts for the counted waits; nobody uses the data).
  const int k_rs = (int)p.k_rs, vt_rs = (int)p.vt_rs;
  // DMA as global_load_lds with a SCALAR tile base and a per-lane byte offset that never changes.  (The buffer_load ... lds
  // form this kernel used first saved the last two vector instructions per DMA, but its LDS-DMA did not always complete in
  // issue order -- the counted vmcnt waits below then let a wave read a ring slot that had not been filled: 8 % of the fp8
  // C5 forwards differed from their repeat in a few dozen tokens, always in the first wave of workgroups, whose K rows sit
  // behind cold TLBs; profiles/r2_attention128_q64_flake.txt.  The global form is the one the GEMMs and the 32-query kernel
  // run with counted waits, without a single mismatch.)  Rows past Skv only exist in the last tile: their lanes re-read its
  // last valid row (masked in the softmax anyway; V^T has its zero pad columns up to a multiple of 64).
  int k_vo[4], v_vo[4];
  int k_vo_lim;
  {
    const int row = tid >> 4, slot = (tid & 15) ^ ((tid >> 4) & 15);
    const int vrow = tid >> 3, vslot = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      k_vo[i] = ((row + i * 16) * k_rs + slot * 8) * 2;
      v_vo[i] = ((vrow + i * 32) * vt_rs + vslot * 8) * 2;
    }
    k_vo_lim = ((Skv - 1 - (n_tiles - 1) * KVB) * k_rs + slot * 8) * 2;
  }
  // one DMA instruction of the pair [K(tk), V(tv)]: pieces 0 - 3 the K rounds, 4 - 7 the V^T rounds
  auto stage_piece = [&](int tk, int tv, auto piece_c) {
    constexpr int PC = decltype(piece_c)::value;
    if constexpr (PC < 4) {
      const int tc = min(tk, n_tiles - 1);
      const char* base = (const char*)(K + (int64_t)tc * KVB * k_rs);
      const int vo = min(k_vo[PC], tc == n_tiles - 1 ? k_vo_lim : 0x7fffffff);
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (uint32_t)vo),
                                       (lptr_t)(k_ring + (tk & (NS - 1)) * K_TILE + (PC * 256 + wave * 64) * 16), 16, 0, 0);
    } else {
      const char* base = (const char*)(VT + (int64_t)min(tv, n_tiles - 1) * KVB);
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (uint32_t)v_vo[PC - 4]),
                                       (lptr_t)(v_ring + (tv & (NS - 1)) * V_TILE + ((PC - 4) * 256 + wave * 64) * 16), 16, 0, 0);
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

  zero_o();
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
  // O^T += V^T P^T over the 32 keys of half-tile (tile, SUB), epilogue form: one fragment at a time through ring slot 0
  auto pv = [&](int tile, auto sub_c, const bf16x8 (&pf)[2][2]) {
    constexpr int SUB = decltype(sub_c)::value;
    const uint32_t vb = (tile & (NS - 1)) * V_TILE;
    auto one = [&](auto k2_c, auto dt_c) {
      constexpr int k2 = decltype(k2_c)::value, dt = decltype(dt_c)::value;
      frag_read<0, dt * 4096>(vc[2 * SUB + k2] + vb);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pv_mfma<dt, 0>(pf[0][k2]);
      pv_mfma<4 + dt, 0>(pf[1][k2]);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    one(I0{}, I0{}); one(I0{}, I1{}); one(I0{}, I2{}); one(I0{}, I3{});
    one(I1{}, I0{}); one(I1{}, I1{}); one(I1{}, I2{}); one(I1{}, I3{});
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
    rescale_o(alpha, qh);
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
    if (__any(!(psum[0] < ALG_Q64_SUM_LIMIT) || !(psum[1] < ALG_Q64_SUM_LIMIT))) {   // 2^40; also inf / NaN -- rare: one branch
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
        if (__any(!(psum[qh] < ALG_Q64_SUM_LIMIT))) fixup(qh, s[qh], pf[qh], psum[qh]);
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];
  };
  // tile boundary, at the top of the EVEN half-tile u = 2 t: K(t+1) and V(t) have landed (one DMA group stays in flight) --
  // a region late for its own reads (K(t) sub 1, V(t-1) sub 1), but the fragment prefetch at the end of this region already
  // reaches into K(t+1).  Every wave is past its reads of K(t-1) and V(t-2) (each was consumed by an MFMA behind a counted
  // wait), so after the barrier their slots take K(t+3) and V(t+2).
  auto boundary = [&](int t) {
    if constexpr (MODE == 2 || MODE == 4)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
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
        pv_mfma<dt, ST & 3>(pp[0][k2]);
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
#ifdef ALG_Q64_NO_FMA   // timing experiment: what a pre-scaled Q with a zero offset would save
      const float a0 = sc[qh][8 * g + 2 * jj];
      const float a1 = sc[qh][8 * g + 2 * jj + 1];
#else
      const float a0 = sc[qh][8 * g + 2 * jj] * c - mc[qh];
      const float a1 = sc[qh][8 * g + 2 * jj + 1] * c - mc[qh];
#endif
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
        pv_mfma<4 + dt, ST & 3>(pp[1][k2]);
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
#ifndef ALG_Q64_ROWSUM_ADD   // (default: the dot2 form)
        psum[qh] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk[qh][g].w[jj]), __builtin_bit_cast(bf2v, 0x3f803f80u), psum[qh], false);
#else   // experiment (round 3): plain fp32 adds -- v_dot2c costs +7 ns per MFMA in an MFMA's shadow (scripts/micro/attn_mix.hip), but
        // this kernel did not get faster with them (1051 vs 1098 TFLOP/s for the default) and the d = 64 form returned NaN: not adopted
        psum[qh] += p0 + p1;
#endif
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
  // the whole prologue is drained (one DMA latency per ~2 ms workgroup): cheap insurance for the first tiles, whose rows
  // sit behind cold TLBs in the first wave of workgroups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  // ROOT CAUSE of the round-2..4 "first round of workgroups" mismatches (profiles/r4_attention128_q64_probe.txt): the S of the
  // half-tile past the end is never used, so hipcc treated the destination registers of the asm MFMAs that compute it as free
  // and recycled them as TEMPORARIES of the softmax right behind those MFMAs -- which write them 32+ cycles after issue (the
  // compiler cannot see an MFMA inside asm text).  An instruction-cache miss between `v_fma` (a0 = s c - m c into the recycled
  // register) and `v_exp` let the MFMA's write land in between: exp2 of a raw score accumulator entered the row sum of the
  // LAST half-tile, whose keys are all masked and whose V^T pad columns are zero -- l inflated, O untouched: whole output rows
  // scaled by 1 / (1 + 2^garbage / l), only the query half whose softmax runs in steps 8 - 15, only where the code was not
  // cached yet.  Keeping the dropped S alive to the end of the region removes the reuse.
  asm volatile("" ::"v"(se[0]), "v"(se[1]));
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  pv(n_tiles - 1, S1{}, po);

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs wrote O a few cycles ago
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    float ot[4][16];
    if (qh == 0) {
      read_o<0>(ot[0]); read_o<1>(ot[1]); read_o<2>(ot[2]); read_o<3>(ot[3]);
    } else {
      read_o<4>(ot[0]); read_o<5>(ot[1]); read_o<6>(ot[2]); read_o<7>(ot[3]);
    }
    const float l_tot = l_run[qh] + __shfl_xor(l_run[qh], 32, 64);
    const float inv = 1.0f / l_tot;
    const int q_row = q_row0 + qh * 32;
#ifdef ALG_EXPERIMENTS
    if (p.dbg && q_row < Sq) {
      float* dp = p.dbg + (((int64_t)b * p.heads + h) * Sq + q_row) * 8 + h2 * 4;   // both lanes of a query: their own partial sums
      dp[0] = l_run[qh], dp[1] = m_run[qh], dp[2] = l_tot, dp[3] = inv;   // (nothing is tracked inside the loop: its code stays as it was)
    }
#endif
    if (q_row < Sq) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 128;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * h2;
          uint2 v;
          v.x = pack_bf2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv);
          v.y = pack_bf2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
  if constexpr (MODE == 3 || MODE == 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tap && lane == 0) {
    uint64_t* cp = p.clk + (size_t)((blockIdx.x >> 6) % p.clk_slots) * 4;
    cp[0] = tap_c0, cp[1] = tap_r0, cp[2] = __builtin_readcyclecounter(), cp[3] = wall_clock64();
  }
}

}  // namespace a128q

#ifdef ALG_EXPERIMENTS
static float* g_q64_tap = nullptr;
#endif
// Returns ALG_OK when launched, 1 when this call is not covered (the caller goes on to attention128_pipe.hip / attention128.hip).
int flash_attn_d128_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                        int64_t o_rs, float scale, hipStream_t stream) {
  using namespace a128q;
  // ALG_ATTN128_Q64: 1 (default) = calls over at least POLICY_TILES KV tiles; 2 = every call the kernel can take; 0 = off.
  // (EXPERIMENTS build: 12 / 13 / 14 = round 3's diagnostic arms, every call the kernel can take.)
  const int enabled = opt(OPT_ATTN128_Q64);
  const int n_tiles = (Skv + KVB - 1) / KVB;
  if (!enabled || n_tiles < (enabled == 1 ? POLICY_TILES : MIN_TILES)) return 1;
  // 31-bit BYTE offsets inside one (batch, head) for the DMA's lane offsets; V^T rows cover whole 64-key tiles
  if ((int64_t)(Skv + 64) * k_rs * 2 >= (1ll << 31) || (int64_t)129 * vt_rs * 2 >= (1ll << 31)) return 1;
  if (vt_rs < (int64_t)((Skv + KVB - 1) / KVB) * KVB) return 1;
  static PerDeviceOnce attr_set;
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
#ifdef ALG_EXPERIMENTS
    for (const void* fn : {(const void*)flash_attn_d128_q64_kernel<1>, (const void*)flash_attn_d128_q64_kernel<2>,
                           (const void*)flash_attn_d128_q64_kernel<3>, (const void*)flash_attn_d128_q64_kernel<4>})
#else
    for (const void* fn : {(const void*)flash_attn_d128_q64_kernel<1>})
#endif
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 1;
    device_mark(attr_set, dev_slot);
  }
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + NW * QW - 1) / (NW * QW);
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.clk = g_clock_tap.load(std::memory_order_acquire);
  p.clk_slots = p.clk ? g_clock_tap_slots.load(std::memory_order_relaxed) : 0;
  if (p.clk_slots <= 0) p.clk = nullptr;
#ifdef ALG_EXPERIMENTS
  p.dbg = g_q64_tap;
#endif
  const int64_t grid = (int64_t)((batch * heads + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) return 1;
  const dim3 g((unsigned)grid), blk(NW * 64);
  switch (enabled) {
#ifdef ALG_EXPERIMENTS
    case 12: hipLaunchKernelGGL(flash_attn_d128_q64_kernel<2>, g, blk, LDS_BYTES, stream, p); break;
    case 13: hipLaunchKernelGGL(flash_attn_d128_q64_kernel<3>, g, blk, LDS_BYTES, stream, p); break;
    case 14: hipLaunchKernelGGL(flash_attn_d128_q64_kernel<4>, g, blk, LDS_BYTES, stream, p); break;
#endif
    default: hipLaunchKernelGGL(flash_attn_d128_q64_kernel<1>, g, blk, LDS_BYTES, stream, p); break;
  }
  return check_launch("alg_flash_attn_d128");
}

}  // namespace alg

#ifdef ALG_EXPERIMENTS
// EXPERIMENTS build only: where the 64-query kernel's launches write their per-query softmax state ([batch][heads][Sq][2 lanes][4]
// float32: partial row sum, running max, pair row sum, its reciprocal), or NULL (default) for nothing.
extern "C" void alg_debug_q64_tap(float* buffer) { alg::g_q64_tap = buffer; }
#endif
