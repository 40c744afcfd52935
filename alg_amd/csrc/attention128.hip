// Flash attention forward, head_dim 128, bf16 in/out, fp32 accumulate, separate query / key lengths: the self-attention
// (32,760 x 32,760 tokens at Wan-480p) and the two cross-attentions (32,760 x 512 text, x 257 image tokens) of the Wan
// DiT.
//
// Same wave-level formulation as attention.hip (d = 64):  S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so a lane owns ONE
// query column and the softmax state is per-lane; P stays in registers as the B operand of O^T = V^T P^T; V^T comes from
// its producer GEMM with kv index bits 2 and 3 swapped.  What changes at d = 128:
//   * 32 MFMAs per 64-row KV tile and wave (16 for S^T over 8 k-steps, 16 for the four 32-row d-tiles of O^T) against
//     the same ~33 v_exp_f32 -- twice the matrix work per softmax instruction, i.e. half the VALU/LDS energy per FLOP
//     (the d = 64 kernel is bound by energy under the package power cap);
//   * a K tile row is 256 B = 16 sixteen-byte slots = all 64 LDS banks: the swizzle XORs the slot with (row & 15), which
//     makes every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) touch 16 distinct slots;
//   * the ragged last KV tile runs in a peeled, masked copy of the loop body (no per-tile v_cndmask), the half-wave max
//     exchange is a v_permlane32_swap, the fma / row sums are packed.
// Workgroup = 8 waves x 32 queries; K / V^T tiles of 64 kv rows stream through a 2-slot LDS ring (2 x 32 KiB) with
// 16-byte global_load_lds; workgroups are ordered so one XCD works on one (batch, head) at a time.
#include "common.h"

namespace alg {

namespace a128 {

constexpr int NW = 8;
constexpr int KVB = 64;
constexpr int K_TILE = KVB * 128 * 2;   // 16 KiB
constexpr int V_TILE = 128 * KVB * 2;   // 16 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2p __attribute__((ext_vector_type(2)));

struct P {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int batch, heads, Sq, Skv, q_blocks;
  int64_t q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs;
  float scale_log2;
  int kv_group;   // query heads per key / value head (grouped-query attention: Llama-3 32 / 8 = 4); 1 = ordinary heads
  // DUAL only: a second key / value set attended by the same queries, o = bf16(bf16(attn(q, k, vt)) + bf16(attn(q, k2, vt2)))
  const bf16_t* k2;
  const bf16_t* vt2;
  int Skv2;
  int64_t k2_bs, k2_rs, vt2_bs, vt2_rs;
};

template <bool B>
struct BoolC { static constexpr bool value = B; };

// CAUSAL: query row i sees keys 0 .. i (the decoder-only language model inside HunyuanVideo's prompt encoder, hy:282-420);
// KV tiles entirely above the workgroup's last query are skipped, tiles that reach past a wave's first query are masked.
// DUAL: the same queries attend two key / value sets one after the other and the two (bf16-rounded) results are added -- the
// image + text cross-attention of the Wan I2V DiT (diffusers WanAttnProcessor2_0: sdpa(q, k_img, v_img) + sdpa(q, k, v)) as ONE
// launch: Q is read once, one output is written, and the separate add kernel (three more passes over [S, 5120]) is gone.  Bit
// for bit what two launches + alg_lincomb give: each set runs the unchanged tile loop from a fresh softmax state.
template <bool CAUSAL, bool DUAL = false>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_d128_kernel(const P p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (K_TILE + V_TILE)];
  char* const k_ring = smem;
  char* const v_ring = smem + 2 * K_TILE;
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
  const int Sq = p.Sq;
  int Skv = p.Skv;
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 128;
  const int hk = h / p.kv_group;
  const bf16_t* K = p.k + (int64_t)b * p.k_bs + hk * 128;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)hk * 128 * p.vt_rs;
  int64_t k_rs = p.k_rs, vt_rs = p.vt_rs;

  // Q^T fragments (B operand): lane (q = l31, h2) holds Q[q][16 ks + 8 h2 .. +8], ks = 0..7
  const int q_row = qb * (NW * 32) + wave * 32 + l31;
  bf16x8 qf[8];
  {
    const bf16_t* qp = Q + (int64_t)min(q_row, Sq - 1) * p.q_rs + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // DMA sources.  K tile: 64 rows x 16 slots, two rounds of 32 rows; physical slot tid & 15 holds logical slot
  // (tid & 15) ^ (row & 15).  V^T tile: 128 d-rows x 8 slots, two rounds of 64 rows, swizzle (row >> 1) & 7.
  const int k_row = tid >> 4;                                  // + 32 per round
  const int k_slot = (tid & 15) ^ ((tid >> 4) & 15);
  const int v_row = tid >> 3;                                  // + 64 per round
  const int v_slot = (tid & 7) ^ ((tid >> 4) & 7);
  const bf16_t* v_src0 = VT + (int64_t)v_row * vt_rs + v_slot * 8;
  const bf16_t* v_src1 = v_src0 + (int64_t)64 * vt_rs;
  auto stage = [&](int slot, int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* ks = K + (int64_t)min(kv0 + k_row + i * 32, Skv - 1) * k_rs + k_slot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(k_ring + slot * K_TILE + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)(v_src0 + kv0), (lptr_t)(v_ring + slot * V_TILE + (wave * 64) * 16), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(v_src1 + kv0), (lptr_t)(v_ring + slot * V_TILE + (512 + wave * 64) * 16), 16,
                                     0, 0);
  };

  const int k_row_off = l31 * 256, k_sw = l31 & 15;
  const int v_row_off = l31 * 128, v_sw = (l31 >> 1) & 7;

  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[i][e] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float c = p.scale_log2;
  int n_tiles = (Skv + KVB - 1) / KVB;
  bool ragged = (Skv & (KVB - 1)) != 0;
  if (CAUSAL) n_tiles = min(n_tiles, (min(qb * (NW * 32) + NW * 32, Sq) + KVB - 1) / KVB);   // keys <= the block's last query
  uint2 o1[DUAL ? 16 : 1];     // DUAL: the first set's output, rounded to bf16 as the single launch stores it

  auto tile = [&](int t, auto masked) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < n_tiles) stage((t + 1) & 1, (t + 1) * KVB);
    const char* Ks = k_ring + (t & 1) * K_TILE + k_row_off;
    const char* Vs = v_ring + (t & 1) * V_TILE + v_row_off;
    // ---- S^T = K Q^T: two 32-row sub-tiles x 8 k-steps ----
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const bf16x8 kf = *(const bf16x8*)(Ks + sub * 8192 + (((2 * ks + h2) ^ k_sw) * 16));
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[sub], 0, 0, 0);
      }
    if (masked.value) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kv = t * KVB + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
          if (kv >= Skv || (CAUSAL && kv > q_row)) s[sub][e] = -INFINITY;
        }
    }
    // ---- online softmax with a LAZY running max (see attention.hip softmax_tile_lazy): probabilities are formed against
    // the current m; the exact tile max / rescale path runs only when a row sum leaves [0, 2^80) (inf on the first tile) ----
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
      for (int dt = 0; dt < 4; ++dt) o_acc[dt] *= alpha;
      psum = probs(m_run * c);
    }
    l_run += psum;
    // ---- O^T += V^T P^T: four 32-row d-tiles x 4 kv blocks ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = *(const bf16x8*)(Vs + dt * 4096 + (((2 * kk + h2) ^ v_sw) * 16));
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], o_acc[dt], 0, 0, 0);
      }
  };
  for (int set = 0; set < (DUAL ? 2 : 1); ++set) {
    if (DUAL && set == 1) {
      // the second key / value set: fresh softmax state, the ring is free once every wave has left the first set's last tile
      const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
      const float inv = 1.0f / l_tot;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          o1[dt * 4 + g].x = pack_bf2(o_acc[dt][4 * g] * inv, o_acc[dt][4 * g + 1] * inv);
          o1[dt * 4 + g].y = pack_bf2(o_acc[dt][4 * g + 2] * inv, o_acc[dt][4 * g + 3] * inv);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[i][e] = 0.0f;
      m_run = -INFINITY;
      l_run = 0.0f;
      Skv = p.Skv2;
      k_rs = p.k2_rs;
      vt_rs = p.vt2_rs;
      K = p.k2 + (int64_t)b * p.k2_bs + hk * 128;
      VT = p.vt2 + (int64_t)b * p.vt2_bs + (int64_t)hk * 128 * vt_rs;
      v_src0 = VT + (int64_t)v_row * vt_rs + v_slot * 8;
      v_src1 = v_src0 + (int64_t)64 * vt_rs;
      n_tiles = (Skv + KVB - 1) / KVB;
      ragged = (Skv & (KVB - 1)) != 0;
      __syncthreads();
    }
    stage(0, 0);
    if (CAUSAL) {
      const int first_masked = (qb * (NW * 32) + wave * 32) / KVB;     // first tile that reaches past this wave's first query
      for (int t = 0; t < n_tiles; ++t) {                               // (wave-uniform; the barrier inside is hit by all)
        if (t < first_masked && !(ragged && t == (Skv + KVB - 1) / KVB - 1))
          tile(t, BoolC<false>{});
        else
          tile(t, BoolC<true>{});
      }
    } else {
      const int n_loop = ragged ? n_tiles - 1 : n_tiles;
      for (int t = 0; t < n_loop; ++t) tile(t, BoolC<false>{});
      if (ragged) tile(n_tiles - 1, BoolC<true>{});
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_row < Sq) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + h * 128;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * h2;
        uint2 v;
        v.x = pack_bf2(o_acc[dt][4 * g] * inv, o_acc[dt][4 * g + 1] * inv);
        v.y = pack_bf2(o_acc[dt][4 * g + 2] * inv, o_acc[dt][4 * g + 3] * inv);
        if (DUAL) {   // bf16 + bf16 in fp32, rounded once: the eager add of the two attention outputs
          const uint2 a = o1[dt * 4 + g];
          auto lo = [](uint32_t u) { return __uint_as_float(u << 16); };
          auto hi = [](uint32_t u) { return __uint_as_float(u & 0xffff0000u); };
          v.x = pack_bf2(lo(a.x) + lo(v.x), hi(a.x) + hi(v.x));
          v.y = pack_bf2(lo(a.y) + lo(v.y), hi(a.y) + hi(v.y));
        }
        *(uint2*)(op + d) = v;
      }
  }
}

}  // namespace a128
}  // namespace alg

using namespace alg;

namespace alg {
int flash_attn_d128_q64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                        int64_t o_rs, float scale, hipStream_t stream);
int flash_attn_d128_pipe(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                         int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t vt_bs, int64_t vt_rs, int64_t o_bs,
                         int64_t o_rs, float scale, hipStream_t stream);
}

static int flash_attn_d128_entry(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq,
                                 int Skv, int64_t q_bstride, int64_t q_rstride, int64_t k_bstride, int64_t k_rstride,
                                 int64_t vt_bstride, int64_t vt_rstride, int64_t o_bstride, int64_t o_rstride,
                                 float scale, int kv_group, int causal, void* stream) {
  if (kv_group <= 0 || heads % kv_group) {
    set_error("alg_flash_attn_d128_ex: heads=%d must be a multiple of kv_group=%d", heads, kv_group);
    return ALG_EINVAL;
  }
  if (causal && Sq != Skv) {
    set_error("alg_flash_attn_d128_ex: causal attention needs Sq == Skv (got %d, %d)", Sq, Skv);
    return ALG_EINVAL;
  }
  if (!q || !k || !vt || !o || batch <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0) {
    set_error("alg_flash_attn_d128: bad argument (batch=%d heads=%d Sq=%d Skv=%d)", batch, heads, Sq, Skv);
    return ALG_EINVAL;
  }
  if (q_rstride % 8 || q_bstride % 8 || k_rstride % 8 || k_bstride % 8 || vt_rstride % 8 || vt_bstride % 8 ||
      o_rstride % 4 || o_bstride % 4 || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) ||
      ((uintptr_t)o & 7)) {
    set_error("alg_flash_attn_d128: q/k/vt need 16-byte aligned rows (strides %% 8 == 0), o 8-byte aligned");
    return ALG_EINVAL;
  }
  if (vt_rstride < (int64_t)((Skv + a128::KVB - 1) / a128::KVB) * a128::KVB) {
    set_error("alg_flash_attn_d128: vt row stride %lld must cover Skv rounded up to %d", (long long)vt_rstride,
              a128::KVB);
    return ALG_EINVAL;
  }
  if (!causal && kv_group == 1) {   // 4,096 keys and more: 64 queries per wave (attention128_q64.hip; default since round 4), 1 = not covered
    const int rc = flash_attn_d128_q64(q, k, vt, o, batch, heads, Sq, Skv, q_bstride, q_rstride, k_bstride, k_rstride, vt_bstride,
                                       vt_rstride, o_bstride, o_rstride, scale, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  if (!causal && kv_group == 1) {   // shorter: the pipelined 32-query kernel (attention128_pipe.hip), 1 = not covered
    const int rp = flash_attn_d128_pipe(q, k, vt, o, batch, heads, Sq, Skv, q_bstride, q_rstride, k_bstride, k_rstride, vt_bstride,
                                        vt_rstride, o_bstride, o_rstride, scale, (hipStream_t)stream);
    if (rp <= 0) return rp;
  }
  a128::P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + a128::NW * 32 - 1) / (a128::NW * 32);
  p.q_bs = q_bstride; p.q_rs = q_rstride; p.k_bs = k_bstride; p.k_rs = k_rstride;
  p.vt_bs = vt_bstride; p.vt_rs = vt_rstride; p.o_bs = o_bstride; p.o_rs = o_rstride;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.kv_group = kv_group;
  p.k2 = nullptr; p.vt2 = nullptr; p.Skv2 = 0; p.k2_bs = p.k2_rs = p.vt2_bs = p.vt2_rs = 0;
  const int nbh = batch * heads;
  const int64_t grid = (int64_t)((nbh + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) {
    set_error("alg_flash_attn_d128: grid too large");
    return ALG_ELIMIT;
  }
  if (causal)
    hipLaunchKernelGGL(a128::flash_attn_d128_kernel<true>, dim3((unsigned)grid), dim3(a128::NW * 64), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(a128::flash_attn_d128_kernel<false>, dim3((unsigned)grid), dim3(a128::NW * 64), 0, (hipStream_t)stream, p);
  return check_launch("alg_flash_attn_d128");
}

// Two key / value sets, one launch (the DUAL form of the kernel above); both sets are short by construction (encoder tokens)
extern "C" int alg_flash_attn_d128_dual(const void* q, const void* k, const void* vt, int Skv, int64_t k_bstride, int64_t k_rstride,
                                        int64_t vt_bstride, int64_t vt_rstride, const void* k2, const void* vt2, int Skv2,
                                        int64_t k2_bstride, int64_t k2_rstride, int64_t vt2_bstride, int64_t vt2_rstride, void* o,
                                        int batch, int heads, int Sq, int64_t q_bstride, int64_t q_rstride, int64_t o_bstride,
                                        int64_t o_rstride, float scale, void* stream) {
  if (!q || !k || !vt || !k2 || !vt2 || !o || batch <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0 || Skv2 <= 0) {
    set_error("alg_flash_attn_d128_dual: bad argument (batch=%d heads=%d Sq=%d Skv=%d Skv2=%d)", batch, heads, Sq, Skv, Skv2);
    return ALG_EINVAL;
  }
  if (q_rstride % 8 || q_bstride % 8 || k_rstride % 8 || k_bstride % 8 || vt_rstride % 8 || vt_bstride % 8 || k2_rstride % 8 ||
      k2_bstride % 8 || vt2_rstride % 8 || vt2_bstride % 8 || o_rstride % 4 || o_bstride % 4 || ((uintptr_t)q & 15) ||
      ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)k2 & 15) || ((uintptr_t)vt2 & 15) || ((uintptr_t)o & 7)) {
    set_error("alg_flash_attn_d128_dual: q/k/vt need 16-byte aligned rows (strides %% 8 == 0), o 8-byte aligned");
    return ALG_EINVAL;
  }
  if (vt_rstride < (int64_t)((Skv + a128::KVB - 1) / a128::KVB) * a128::KVB ||
      vt2_rstride < (int64_t)((Skv2 + a128::KVB - 1) / a128::KVB) * a128::KVB) {
    set_error("alg_flash_attn_d128_dual: vt row strides %lld / %lld must cover Skv / Skv2 rounded up to %d", (long long)vt_rstride,
              (long long)vt2_rstride, a128::KVB);
    return ALG_EINVAL;
  }
  a128::P p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.Sq = Sq; p.Skv = Skv;
  p.q_blocks = (Sq + a128::NW * 32 - 1) / (a128::NW * 32);
  p.q_bs = q_bstride; p.q_rs = q_rstride; p.k_bs = k_bstride; p.k_rs = k_rstride;
  p.vt_bs = vt_bstride; p.vt_rs = vt_rstride; p.o_bs = o_bstride; p.o_rs = o_rstride;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.kv_group = 1;
  p.k2 = (const bf16_t*)k2; p.vt2 = (const bf16_t*)vt2; p.Skv2 = Skv2;
  p.k2_bs = k2_bstride; p.k2_rs = k2_rstride; p.vt2_bs = vt2_bstride; p.vt2_rs = vt2_rstride;
  const int nbh = batch * heads;
  const int64_t grid = (int64_t)((nbh + 7) / 8) * 8 * p.q_blocks;
  if (grid > 0x7fffffff) {
    set_error("alg_flash_attn_d128_dual: grid too large");
    return ALG_ELIMIT;
  }
  hipLaunchKernelGGL((a128::flash_attn_d128_kernel<false, true>), dim3((unsigned)grid), dim3(a128::NW * 64), 0, (hipStream_t)stream, p);
  return check_launch("alg_flash_attn_d128_dual");
}

extern "C" int alg_flash_attn_d128(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq,
                                   int Skv, int64_t q_bstride, int64_t q_rstride, int64_t k_bstride, int64_t k_rstride,
                                   int64_t vt_bstride, int64_t vt_rstride, int64_t o_bstride, int64_t o_rstride,
                                   float scale, void* stream) {
  return flash_attn_d128_entry(q, k, vt, o, batch, heads, Sq, Skv, q_bstride, q_rstride, k_bstride, k_rstride, vt_bstride,
                               vt_rstride, o_bstride, o_rstride, scale, 1, 0, stream);
}

extern "C" int alg_flash_attn_d128_ex(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq,
                                      int Skv, int64_t q_bstride, int64_t q_rstride, int64_t k_bstride, int64_t k_rstride,
                                      int64_t vt_bstride, int64_t vt_rstride, int64_t o_bstride, int64_t o_rstride,
                                      float scale, int kv_group, int causal, void* stream) {
  return flash_attn_d128_entry(q, k, vt, o, batch, heads, Sq, Skv, q_bstride, q_rstride, k_bstride, k_rstride, vt_bstride,
                               vt_rstride, o_bstride, o_rstride, scale, kv_group, causal, stream);
}
