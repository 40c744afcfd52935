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
// Workgroup = 8 waves x 32 queries = 256 queries of one (batch, head); KV tiles of 64 stream through a
// double-buffered 32 KiB LDS ring with 16-byte global_load_lds; the bank swizzle sits on the source address
// (same scheme as gemm.hip).  Workgroups are ordered so that one XCD works on one head at a time (K/V of a
// head = 4.5 MB, re-read by the 70 query blocks of that head out of the XCD's L2).
#include "common.h"

namespace alg {

constexpr int ATT_THREADS = 512;
constexpr int QB = 256;   // queries per workgroup
constexpr int KVB = 64;   // kv rows per tile
constexpr int ATT_TILE = KVB * 64 * 2;     // 8 KiB (K tile; V^T tile is the same size)
constexpr int ATT_LDS = 4 * ATT_TILE;      // 2 stages x (K + V^T)

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
};

__global__ __launch_bounds__(ATT_THREADS) void flash_attn_d64_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[ATT_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h2 = lane >> 5;

  // ---- workgroup -> (batch*head, q block): XCD x walks heads x, x+8, ... ----
  const int nbh = p.batch * p.heads;
  int bh, qb;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int per = (nbh + 7) >> 3;               // (batch*head) slots per XCD
    const int slot = idx / p.q_blocks;
    qb = idx - slot * p.q_blocks;
    bh = slot * 8 + xcd;
    if (slot >= per || bh >= nbh) return;
  }
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int S = p.S;
  const bf16_t* Q = p.q + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* K = p.k + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* VT = p.vt + (int64_t)b * p.vt_bs + (int64_t)h * 64 * p.vt_rs;

  // ---- Q^T fragments (B operand): lane (q = l31, h2) holds Q[q][16 ks + 8 h2 .. +8] ----
  const int q_row = qb * QB + wave * 32 + l31;
  bf16x8 qf[4];
  {
    const bf16_t* qp = Q + (int64_t)min(q_row, S - 1) * p.q_rs + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- DMA sources: one 16-B piece of K and one of V^T per thread per tile ----
  const int srow = tid >> 3;                        // K: kv row, V^T: d row   (0..63)
  const int sslot = (tid & 7) ^ ((tid >> 4) & 7);
  const bf16_t* vt_src = VT + (int64_t)srow * p.vt_rs + sslot * 8;
  auto stage = [&](int buf, int kv0) {
    char* base = smem + buf * 2 * ATT_TILE;
    const bf16_t* ks = K + (int64_t)min(kv0 + srow, S - 1) * p.q_rs + sslot * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(base + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vt_src + kv0), (lptr_t)(base + ATT_TILE + wave * 1024), 16, 0, 0);
  };

  const int sw = (l31 >> 1) & 7;
  const int frag_row_off = l31 * 128;   // + sub*32*128 (K) / dt*32*128 (V^T)

  f32x16 o_acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[i][e] = 0.0f;
  float m_run = -INFINITY;   // running max of raw scores (this lane's query)
  float l_run = 0.0f;        // this half-wave's share of the running sum
  const float c = p.scale_log2;

  const int n_tiles = (S + KVB - 1) / KVB;
  stage(0, 0);
  for (int t = 0; t < n_tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < n_tiles) stage((t + 1) & 1, (t + 1) * KVB);
    const char* Ks = smem + (t & 1) * 2 * ATT_TILE;
    const char* Vs = Ks + ATT_TILE;

    // ---- S^T = K Q^T : two 32-kv sub-tiles ----
    f32x16 s_acc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s_acc[sub][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Ks + frag_row_off + sub * 4096 + (((2 * ks + h2) ^ sw) * 16));
        s_acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_acc[sub], 0, 0, 0);
      }
    }
    // mask kv >= S (last tile only)
    if (t == n_tiles - 1 && (S & (KVB - 1))) {
      const int kv0 = t * KVB;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kv = kv0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
          if (kv >= S) s_acc[sub][e] = -INFINITY;
        }
    }
    // ---- online softmax (per-lane scalars) ----
    float mt = s_acc[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mt = fmaxf(mt, s_acc[sub][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    m_run = m_new;
    float psum = 0.0f;
    bf16x8 pf[4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(s_acc[sub][8 * g + 2 * j] * c - mc);
          const float p1 = __builtin_amdgcn_exp2f(s_acc[sub][8 * g + 2 * j + 1] * c - mc);
          // the row sum uses the unrounded fp32 probabilities (as the math SDPA path does)
          psum += p0 + p1;
          pk.u[j] = pack_bf2(p0, p1);
        }
        pf[sub * 2 + g] = pk.v;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {   // kk = sub*2 + g : kv block [16 kk, 16 kk + 16)
        const bf16x8 vf = *(const bf16x8*)(Vs + frag_row_off + dt * 4096 + (((2 * kk + h2) ^ sw) * 16));
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], o_acc[dt], 0, 0, 0);
      }
  }

  // ---- finish: combine the half-waves' sums, normalise, store O[q][d] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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

}  // namespace alg

using namespace alg;

extern "C" int alg_flash_attn_d64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S,
                                  int64_t q_bstride, int64_t q_rstride, int64_t vt_bstride, int64_t vt_rstride,
                                  int64_t o_bstride, int64_t o_rstride, float scale, void* stream) {
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
  AttnP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.batch = batch; p.heads = heads; p.S = S;
  p.q_blocks = (S + QB - 1) / QB;
  p.q_bs = q_bstride; p.q_rs = q_rstride; p.vt_bs = vt_bstride; p.vt_rs = vt_rstride;
  p.o_bs = o_bstride; p.o_rs = o_rstride;
  p.scale_log2 = scale * 1.4426950408889634f;
  const int nbh = batch * heads;
  const int64_t grid = (int64_t)((nbh + 7) / 8) * 8 * p.q_blocks;
  hipLaunchKernelGGL(flash_attn_d64_kernel, dim3((unsigned)grid), dim3(ATT_THREADS), 0, (hipStream_t)stream, p);
  return check_launch("alg_flash_attn_d64");
}
