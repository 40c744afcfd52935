// Flash attention forward, head_dim 64, the long self-attention form of the CogVideoX DiT: 64 queries per wave, one wave per
// SIMD -- the d = 64 sibling of attention128_q64.hip (read its header for the reasoning; this file only states what differs).
//
//   * A 32-key half-tile is 16 MFMAs (8 for S^T over four k-steps and two query halves, 8 for the two 32-row d-tiles of O^T over
//     two kv blocks and two query halves) against the SAME 32 x 64 scores as at d = 128: 4 VALU instructions per MFMA in the
//     common path.  That only fits because the scores arrive in log2 units (Q pre-scaled where it is produced,
//     ALG_ATTN_Q_PRESCALED) and the running offset is snapped to zero on the first tile (attention.hip, softmax_tile_zero):
//     p = exp2(s) with no subtraction.  A region runs in the ZERO form when every lane's offset is 0, in the subtracting form
//     otherwise.  Only pre-scaled calls take this kernel.
//   * K and V^T tiles are both 64 rows x 128 bytes (8 KiB, swizzle (row >> 1) & 7): two 4-slot rings = 64 KiB of LDS.
//   * Eight steps per region: {K k-step, PV block} x 4, two score pairs of the softmax per step; four DMA pieces per tile.
//   * O = a[0:63] (tile (qh, dt) = a[16 (2 qh + dt) .. + 15]), Q = 32 AccVGPRs, fragment ring a[240:255].
// The split-KV tail of alg_flash_attn_d64 stays on attention.hip's kernel: this one replaces the MAIN launch only (same
// 256-query blocks, same block -> (head, q block) order).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// EXPERIMENTS-only kernel: keeps the row-sum limit it was validated with (2^40; the product kernels moved to 2^80 in round 4 --
// the d = 64 sibling returned wrong rows at 2,050 keys with the larger limit, not investigated)
#define ALG_Q64_SUM_LIMIT 1.0995116e12f

namespace alg {
namespace a64q {

constexpr int NW = 4;
constexpr int QW = 64;
constexpr int KVB = 64;
constexpr int TILE = KVB * 64 * 2;       // 8 KiB: K tile = V^T tile
constexpr int NS = 4;
constexpr int LDS_BYTES = 2 * NS * TILE;
constexpr int MIN_TILES = 8;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct P {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int batch, heads, S, q_blocks;
  int64_t q_bs, q_rs, vt_bs, vt_rs, o_bs, o_rs;
};

#define ALG_FRAG_CLOBBER "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", \
                         "a252", "a253", "a254", "a255"
#define ALG_FR0 "a[240:243]"
#define ALG_FR1 "a[244:247]"
#define ALG_FR2 "a[248:251]"
#define ALG_FR3 "a[252:255]"
template <int SLOT, int OFF>
__device__ __forceinline__ void frag_read(uint32_t addr) {
  if constexpr (SLOT == 0) asm volatile("ds_read_b128 " ALG_FR0 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 1) asm volatile("ds_read_b128 " ALG_FR1 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 2) asm volatile("ds_read_b128 " ALG_FR2 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
  if constexpr (SLOT == 3) asm volatile("ds_read_b128 " ALG_FR3 ", %0 offset:%1" ::"v"(addr), "n"(OFF) : ALG_FRAG_CLOBBER);
}
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
// O is NOT a C++ value inside the loop: it lives in a[0:63] (tile (qh, dt) = a[16 (2 qh + dt) .. + 15]) and is named literally by
// every asm that touches it (all of them list a0 - a63 as clobbers).  As an asm OPERAND pinned to those registers hipcc kept O
// in ArchVGPRs between the asms -- 16 v_accvgpr_write in front of every MFMA and reads right behind it, i.e. behind an MFMA
// it cannot see (wrong results, not just slow).
#define ALG_O_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
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
  ALG_PV_CASE(0, 0, 15) ALG_PV_CASE(1, 16, 31) ALG_PV_CASE(2, 32, 47) ALG_PV_CASE(3, 48, 63)
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
                 "s_nop 0"
                 : "=&v"(tmp) : "v"(alpha) : ALG_O_CLOBBER);
  } else {
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
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
}

__global__ __launch_bounds__(NW * 64) void flash_attn_d64_q64_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const k_ring = smem;
  char* const v_ring = smem + NS * TILE;
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
  const int S = p.S;
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* K = p.k + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 64 * p.vt_rs;

  const int q_row0 = qb * (NW * QW) + wave * QW + l31;
  bf16x8 qf[2][4];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const bf16_t* qp = Q + (int64_t)min(q_row0 + qh * 32, S - 1) * p.q_rs + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qh][ks] = *(const bf16x8*)(qp + ks * 16);
  }

  const int n_tiles = (S + KVB - 1) / KVB;
  const bool ragged = (S & (KVB - 1)) != 0;

  // DMA: buffer_load ... lds, tile origin in the scalar offset, constant per-lane byte offsets (rows past S read as zeros).
  // Both tiles: 64 rows x 8 slots, two rounds of 32 rows; physical slot tid & 7 holds logical slot (tid & 7) ^ ((row >> 1) & 7).
  const int q_rs = (int)p.q_rs, vt_rs = (int)p.vt_rs;
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)(((int64_t)(S - 1) * q_rs + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)VT, 0, (int)((int64_t)64 * vt_rs * 2), 0x00020000);
  int k_vo[2], v_vo[2];
  {
    const int row = tid >> 3, slot = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      k_vo[i] = ((row + i * 32) * q_rs + slot * 8) * 2;
      v_vo[i] = ((row + i * 32) * vt_rs + slot * 8) * 2;
    }
  }
  auto stage_piece = [&](int tk, int tv, auto piece_c) {   // pieces 0, 1: the K rounds of tile tk; 2, 3: the V^T rounds of tile tv
    constexpr int PC = decltype(piece_c)::value;
    if constexpr (PC < 2) {
      const int so = min(tk, n_tiles - 1) * KVB * q_rs * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(k_ring + (tk & (NS - 1)) * TILE + (PC * 256 + wave * 64) * 16), 16,
                                               k_vo[PC], so, 0, 0);
    } else {
      const int so = min(tv, n_tiles - 1) * KVB * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(v_ring + (tv & (NS - 1)) * TILE + ((PC - 2) * 256 + wave * 64) * 16), 16,
                                               v_vo[PC - 2], so, 0, 0);
    }
  };
  auto stage_k = [&](int tile) {
    stage_piece(tile, 0, std::integral_constant<int, 0>{});
    stage_piece(tile, 0, std::integral_constant<int, 1>{});
  };
  auto stage_v = [&](int tile) {
    stage_piece(0, tile, std::integral_constant<int, 2>{});
    stage_piece(0, tile, std::integral_constant<int, 3>{});
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const int row_off = l31 * 128, sw = (l31 >> 1) & 7;
  // per-lane fragment addresses without the (slot, half, d-tile) part: kc[k-step], vc[16-key block of the 64-key tile]
  uint32_t kc[4], vc[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kc[ks] = lds0 + row_off + (((2 * ks + h2) ^ sw) * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) vc[j] = lds0 + NS * TILE + row_off + (((2 * j + h2) ^ sw) * 16);

  zero_o();
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};

  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;
  auto qk = [&](int tile, auto sub_c, f32x16 (&s)[2]) {   // prologue form
    constexpr int SUB = decltype(sub_c)::value;
    const char* Ks = k_ring + (tile & (NS - 1)) * TILE + row_off + SUB * 4096;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(Ks + (((2 * ks + h2) ^ sw) * 16));
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const bf16x8 qv = qf[qh][ks];
        if (ks == 0)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s[qh]) : "v"(kf), "a"(qv));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[qh]) : "v"(kf), "a"(qv));
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // XDL write -> VALU read hazard, invisible to hipcc
  };
  auto pv = [&](int tile, auto sub_c, const bf16x8 (&pf)[2][2]) {   // epilogue form: one fragment at a time through ring slot 0
    constexpr int SUB = decltype(sub_c)::value;
    const uint32_t vb = (tile & (NS - 1)) * TILE;
    auto one = [&](auto k2_c, auto dt_c) {
      constexpr int k2 = decltype(k2_c)::value, dt = decltype(dt_c)::value;
      frag_read<0, dt * 4096>(vc[2 * SUB + k2] + vb);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pv_mfma<dt, 0>(pf[0][k2]);
      pv_mfma<2 + dt, 0>(pf[1][k2]);
    };
    one(S0_{}, S0_{}); one(S0_{}, S1_{}); one(S1_{}, S0_{}); one(S1_{}, S1_{});
  };
  typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
  auto probs = [&](const f32x16& s, float m, bf16x8 (&pf)[2]) -> float {   // scores are in log2 units
    float psum = 0.0f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p0 = __builtin_amdgcn_exp2f(s[8 * g + 2 * j] - m);
        const float p1 = __builtin_amdgcn_exp2f(s[8 * g + 2 * j + 1] - m);
        pk.u[j] = pack_bf2(p0, p1);
        psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk.u[j]), __builtin_bit_cast(bf2v, 0x3f803f80u), psum,
                                              false);
      }
      pf[g] = pk.v;
    }
    return psum;
  };
  // exact path of the lazy running max; on the first tile the offset is snapped to zero when that is safe (|max| < 64)
  auto fixup = [&](int qh, const f32x16& s, bf16x8 (&pf)[2], float& psum) {
    float mt = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    float m_new = fmaxf(m_run[qh], mt);
    if (m_new == -INFINITY) {
      psum = 0.0f;
      return;
    }
    if (m_run[qh] == -INFINITY && fabsf(m_new) < 64.0f) m_new = 0.0f;
    const float alpha = __builtin_amdgcn_exp2f(m_run[qh] - m_new);
    m_run[qh] = m_new;
    l_run[qh] *= alpha;
    rescale_o(alpha, qh);
    psum = probs(s, m_new, pf);
  };
  auto mask_tail = [&](int kv_base, f32x16 (&s)[2]) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kv = kv_base + (e & 3) + 8 * (e >> 2) + 4 * h2;
        if (kv >= S) s[qh][e] = -INFINITY;
      }
  };
  auto finish_softmax = [&](f32x16 (&s)[2], bf16x8 (&pf)[2][2], float (&psum)[2]) {
    if (__any(!(psum[0] < ALG_Q64_SUM_LIMIT) || !(psum[1] < ALG_Q64_SUM_LIMIT))) {
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
        if (__any(!(psum[qh] < ALG_Q64_SUM_LIMIT))) fixup(qh, s[qh], pf[qh], psum[qh]);
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];
  };
  // top of the EVEN half-tile u = 2 t: K(t+1) and V(t) have landed (one DMA group of four stays in flight); K(t+3) and
  // V(t+2) go out piece by piece during the region
  auto boundary = [&]() {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 se[2], so[2];
  bf16x8 pe[2][2], po[2][2];
  // One pipelined half-tile u (CUR = u & 1), eight steps {K k-step | PV block}: S of u + 1 and the PV of u - 1 under the
  // softmax of u.  SL: ring slot of tile t = u >> 1 as a constant, or -1 (runtime).  ZERO: every lane's offset is 0.
  auto region = [&](int u, auto cur_c, auto sl_c, auto zero_c, f32x16 (&sc)[2], f32x16 (&sn)[2], bf16x8 (&pc)[2][2],
                    const bf16x8 (&pp)[2][2]) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int SL = decltype(sl_c)::value;
    constexpr bool ZERO = decltype(zero_c)::value;
    const int t = u >> 1;
    constexpr int KSUB = CUR == 0 ? 1 : 0, VSUB = CUR == 0 ? 1 : 0, VNSUB = CUR;
    const int kt = CUR == 0 ? t : t + 1, vt_ = CUR == 0 ? t - 1 : t;
    const uint32_t ks_off = (kt & (NS - 1)) * TILE + KSUB * 4096;
    const uint32_t vs_off = (vt_ & (NS - 1)) * TILE;
    const uint32_t kn_off = ((t + 1) & (NS - 1)) * TILE + CUR * 4096;
    const uint32_t vn_off = (t & (NS - 1)) * TILE;
    auto rd = [&](auto step_c) {   // fragment of step ST (8 .. 10: steps 0 .. 2 of the next region) -> ring slot ST & 3
      constexpr int ST = decltype(step_c)::value;
      if constexpr (SL >= 0) {
        constexpr int KSL = CUR == 0 ? SL : (SL + 1) & 3, VSL = CUR == 0 ? (SL + 3) & 3 : SL;
        if constexpr (ST >= 8) {
          constexpr int S2 = ST - 8;
          if constexpr ((S2 & 1) == 0) {
            frag_read<ST & 3, ((SL + 1) & 3) * TILE + CUR * 4096>(kc[S2 >> 1]);
          } else {
            frag_read<ST & 3, SL * TILE>(vc[2 * VNSUB]);                       // next region, PV block 0: k2 = 0, dt = 0
          }
        } else if constexpr ((ST & 1) == 0) {
          frag_read<ST & 3, KSL * TILE + KSUB * 4096>(kc[ST >> 1]);
        } else {
          constexpr int k2 = (ST >> 1) >> 1, dt = (ST >> 1) & 1;
          frag_read<ST & 3, VSL * TILE + dt * 4096>(vc[2 * VSUB + k2]);
        }
      } else if constexpr (ST >= 8) {
        constexpr int S2 = ST - 8;
        if constexpr ((S2 & 1) == 0)
          frag_read<ST & 3, 0>(kc[S2 >> 1] + kn_off);
        else
          frag_read<ST & 3, 0>(vc[2 * VNSUB] + vn_off);
      } else if constexpr ((ST & 1) == 0) {
        frag_read<ST & 3, 0>(kc[ST >> 1] + ks_off);
      } else {
        constexpr int k2 = (ST >> 1) >> 1, dt = (ST >> 1) & 1;
        frag_read<ST & 3, dt * 4096>(vc[2 * VSUB + k2] + vs_off);
      }
    };
    float psum[2] = {0.0f, 0.0f};
    union { bf16x8 v; uint32_t w[4]; } pk[2][2];
    auto pair = [&](auto pidx_c) {   // score pair PIDX of the 32 x 64 half-tile: query half, register quad, pair
      constexpr int PIDX = decltype(pidx_c)::value;
      constexpr int qh = PIDX >> 3, g = (PIDX >> 2) & 1, jj = PIDX & 3;
      const float x0 = ZERO ? sc[qh][8 * g + 2 * jj] : sc[qh][8 * g + 2 * jj] - m_run[qh];
      const float x1 = ZERO ? sc[qh][8 * g + 2 * jj + 1] : sc[qh][8 * g + 2 * jj + 1] - m_run[qh];
      const float p0 = __builtin_amdgcn_exp2f(x0), p1 = __builtin_amdgcn_exp2f(x1);
      pk[qh][g].w[jj] = pack_bf2(p0, p1);
#ifndef ALG_Q64_ROWSUM_ADD   // (default: the dot2 form)
      psum[qh] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk[qh][g].w[jj]), __builtin_bit_cast(bf2v, 0x3f803f80u), psum[qh], false);
#else   // experiment (round 3): plain fp32 adds -- v_dot2c costs +7 ns per MFMA in an MFMA's shadow (scripts/micro/attn_mix.hip), but
        // this kernel did not get faster with them (1051 vs 1098 TFLOP/s for the default) and the d = 64 form returned NaN: not adopted
      psum[qh] += p0 + p1;
#endif
    };
    auto step = [&](auto step_c) {
      constexpr int ST = decltype(step_c)::value;
      constexpr bool QK = (ST & 1) == 0;
      constexpr int KS = ST >> 1;
      constexpr int k2 = (ST >> 1) >> 1, dt = (ST >> 1) & 1;
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // reads ST + 1, ST + 2 in flight: fragment ST has arrived
      if constexpr (QK)
        qk_mfma<ST & 3, KS == 0>(sn[0], qf[0][KS]);
      else
        pv_mfma<dt, ST & 3>(pp[0][k2]);
      pair(std::integral_constant<int, 2 * ST>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (QK)
        qk_mfma<ST & 3, KS == 0>(sn[1], qf[1][KS]);
      else
        pv_mfma<2 + dt, ST & 3>(pp[1][k2]);
      rd(std::integral_constant<int, ST + 3>{});
      if constexpr (CUR == 0 && (ST & 1) == 1) stage_piece(t + 3, t + 2, std::integral_constant<int, (ST >> 1)>{});
      pair(std::integral_constant<int, 2 * ST + 1>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
      for (int g = 0; g < 2; ++g) pc[qh][g] = pk[qh][g].v;
    finish_softmax(sc, pc, psum);
  };
  // boundary / tail mask / choice of the softmax form, then the region
  auto half_tile = [&](int u, auto cur_c, auto sl_c, f32x16 (&sc)[2], f32x16 (&sn)[2], bf16x8 (&pc)[2][2],
                       const bf16x8 (&pp)[2][2]) {
    constexpr int CUR = decltype(cur_c)::value;
    const int t = u >> 1;
    if (CUR == 0) boundary();
    if (ragged && t == n_tiles - 1) mask_tail(t * KVB + CUR * 32, sc);
    __builtin_amdgcn_sched_barrier(0);
#ifndef ALG_Q64D64_NO_ZERO
    if (__all(m_run[0] == 0.0f && m_run[1] == 0.0f))
      region(u, cur_c, sl_c, std::true_type{}, sc, sn, pc, pp);
    else
#endif
      region(u, cur_c, sl_c, std::false_type{}, sc, sn, pc, pp);
  };

  // ---- prologue: K(0), K(1), V(0), [K(2), V(1)], [K(3), V(2)]; S of half-tiles 0 and 1, softmax of 0 un-pipelined ----
  stage_k(0);
  stage_k(1);
  stage_v(0);
  stage_k(2);
  stage_v(1);
  stage_k(3);
  stage_v(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the whole prologue drained (kept from the time a cold-start DMA-order race was suspected in attention128_q64.hip; the cause turned out to be elsewhere -- see the end of this kernel)
  __builtin_amdgcn_s_barrier();
  qk(0, S0{}, se);
  {
    float psum[2];
    qk(0, S1{}, so);
    psum[0] = probs(se[0], m_run[0], pe[0]);
    psum[1] = probs(se[1], m_run[1], pe[1]);
    finish_softmax(se, pe, psum);
  }
  asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // K(1), V(0): what half-tile 1 reads
  __builtin_amdgcn_s_barrier();
  frag_read<0, 0>(kc[0] + 1 * TILE);      // region 1, step 0: K(1) sub 0, k-step 0
  frag_read<1, 0>(vc[0] + 0 * TILE);      //           step 1: V(0) sub 0, kv block 0, d-tile 0
  frag_read<2, 0>(kc[1] + 1 * TILE);      //           step 2: K(1) sub 0, k-step 1
  using SLR = std::integral_constant<int, -1>;
  int t = 0;
#ifndef ALG_Q64D64_RUNTIME_SL
  for (; t + 4 <= n_tiles - 1; t += 4) {   // t is a multiple of 4 here: tile t + i sits in slot i
    half_tile(2 * t + 1, S1{}, std::integral_constant<int, 0>{}, so, se, po, pe);
    half_tile(2 * t + 2, S0{}, std::integral_constant<int, 1>{}, se, so, pe, po);
    half_tile(2 * t + 3, S1{}, std::integral_constant<int, 1>{}, so, se, po, pe);
    half_tile(2 * t + 4, S0{}, std::integral_constant<int, 2>{}, se, so, pe, po);
    half_tile(2 * t + 5, S1{}, std::integral_constant<int, 2>{}, so, se, po, pe);
    half_tile(2 * t + 6, S0{}, std::integral_constant<int, 3>{}, se, so, pe, po);
    half_tile(2 * t + 7, S1{}, std::integral_constant<int, 3>{}, so, se, po, pe);
    half_tile(2 * t + 8, S0{}, std::integral_constant<int, 0>{}, se, so, pe, po);
  }
#endif
  for (; t < n_tiles - 1; ++t) {
    half_tile(2 * t + 1, S1{}, SLR{}, so, se, po, pe);
    half_tile(2 * t + 2, S0{}, SLR{}, se, so, pe, po);
  }
  half_tile(2 * n_tiles - 1, S1{}, SLR{}, so, se, po, pe);
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
  float ot[2][2][16];
  read_o<0>(ot[0][0]);
  read_o<1>(ot[0][1]);
  read_o<2>(ot[1][0]);
  read_o<3>(ot[1][1]);
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const float l_tot = l_run[qh] + __shfl_xor(l_run[qh], 32, 64);
    const float inv = 1.0f / l_tot;
    const int q_row = q_row0 + qh * 32;
    if (q_row < S) {
      bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * h2;
          uint2 v;
          v.x = pack_bf2(ot[qh][dt][4 * g] * inv, ot[qh][dt][4 * g + 1] * inv);
          v.y = pack_bf2(ot[qh][dt][4 * g + 2] * inv, ot[qh][dt][4 * g + 3] * inv);
          *(uint2*)(op + d) = v;
        }
    }
  }
}

}  // namespace a64q

// Main launch of alg_flash_attn_d64_ex for pre-scaled Q: `blocks` workgroups in the order of attention.hip's kernel (block ->
// XCD, head slot, q block).  Returns ALG_OK when launched, 1 when this call is not covered.
int flash_attn_d64_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S, int q_blocks,
                       int64_t q_bs, int64_t q_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs, int64_t o_rs, unsigned blocks,
                       hipStream_t stream) {
  using namespace a64q;
  // OPT-IN (ALG_ATTN64_Q64=1): measured 964 vs 1026 TFLOP/s for attention.hip's 8 x 32-query kernel at the C2 shape.  At
  // d = 64 a half-tile offers 16 MFMAs for the same 64 VALU instructions of softmax: 4 per MFMA, about 24 issue cycles of exp2
  // / pack / dot2 against the 28 an MFMA leaves free, before the fragment read, the waits and the DMA -- one in-order wave per
  // SIMD cannot keep the matrix pipe fed, while four waves per SIMD of the 32-query kernel overlap freely.  Kept, tested and
  // bit-compatible, as the A/B reference for that statement.
  if (opt(OPT_ATTN64_Q64) != 1 || (S + KVB - 1) / KVB < MIN_TILES || blocks == 0) return 1;
  if (q_blocks != (S + NW * QW - 1) / (NW * QW)) return 1;
  if ((int64_t)(S + 64) * q_rs * 2 >= (1ll << 31) || (int64_t)65 * vt_rs * 2 >= (1ll << 31)) return 1;   // 31-bit byte offsets
  if (vt_rs < (int64_t)((S + KVB - 1) / KVB) * KVB) return 1;
  static PerDeviceOnce attr_set;
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
    if (hipFuncSetAttribute((const void*)flash_attn_d64_q64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return 1;
    device_mark(attr_set, dev_slot);
  }
  P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.S = S; p.q_blocks = q_blocks;
  p.q_bs = q_bs; p.q_rs = q_rs; p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  hipLaunchKernelGGL(flash_attn_d64_q64_kernel, dim3(blocks), dim3(NW * 64), LDS_BYTES, stream, p);
  return check_launch("alg_flash_attn_d64");
}

}  // namespace alg
