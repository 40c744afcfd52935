// C ABI entry of the bf16 GEMM: argument validation and schedule dispatch.  The kernel template lives in gemm_kernel.h; one
// translation unit per schedule (gemm_p0.hip, gemm_p6.hip, gemm_p7.hip, gemm_p8.hip, gemm_p9.hip) instantiates it.
#include <stdlib.h>

#include "common.h"

namespace alg {
constexpr int BM = 256, BN = 256;
int launch_gemm_p6(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p9(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p9_pair(const alg_gemm_args* a, int m_tiles_a, int n_tiles_a, const alg_gemm_args* b, int m_tiles_b, int n_tiles_b,
                        hipStream_t s);
int launch_gemm_p9_pair_qk(const alg_gemm_args* a, int m_tiles_a, int n_tiles_a, const alg_gemm_args* b, int m_tiles_b, int n_tiles_b,
                           const alg_qk_norm_rope_args* e, hipStream_t s);
int launch_gemm_p10(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p10_pair(const alg_gemm_args* a, int m_tiles_a, int n_tiles_a, const alg_gemm_args* b, int m_tiles_b, int n_tiles_b,
                         hipStream_t s);
int launch_gemm_p10_pair_qk(const alg_gemm_args* a, int m_tiles_a, int n_tiles_a, const alg_gemm_args* b, int m_tiles_b, int n_tiles_b,
                            const alg_qk_norm_rope_args* e, hipStream_t s);
int launch_gemm_p11(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p6_fp8(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p9_fp8(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
int launch_gemm_p6_conv(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s);
// ALG_GEMM_PIPE: 10 (default since round 6: schedule 9's 4-wave asm main loop on v_mfma_f32_16x16x32_bf16, the shape that
// sustains ~10 % more under the package power cap; fp32 rounding points differ from 9 / 6: 32 products per instruction) | 9 (the
// asm main loop on 32x32x16; round 3: faster than the 8-wave ping-pong on all five C2 shapes) | 6 (the 8-wave ping-pong;
// bit-identical to 9).  Calls the asm loops cannot take (K < 128, byte offsets past 32 bits) and the convolution operands run
// schedule 6; e4m3 operands run schedule 9's e4m3 loop under 9 and 10.
static int gemm_pipe() { return opt(OPT_GEMM_PIPE); }
static bool asm_pipe() { return gemm_pipe() == 9 || gemm_pipe() == 10; }
}  // namespace alg

using namespace alg;

static int gemm_entry(const alg_gemm_args* a, void* stream, bool fp8, bool validate_only = false) {
  if (!a || !a->A || !a->B || !a->C) {
    set_error("alg_gemm_bf16: null argument");
    return ALG_EINVAL;
  }
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) {
    set_error("alg_gemm_bf16: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
    return ALG_EINVAL;
  }
  if (a->K % (fp8 ? 128 : 64) != 0) {
    set_error("alg_gemm_%s: K=%d must be a multiple of %d", fp8 ? "fp8" : "bf16", a->K, fp8 ? 128 : 64);
    return ALG_EINVAL;
  }
  const int al = fp8 ? 16 : 8;  // elements per 16 bytes
  if (a->lda % al || a->ldb % al || a->strideA % al || a->strideB % al || ((uintptr_t)a->A & 15) ||
      ((uintptr_t)a->B & 15)) {
    set_error("alg_gemm: A/B must be 16-byte aligned with lda/ldb/strides multiples of 16 bytes");
    return ALG_EINVAL;
  }
  if (fp8 && (!a->a_scale || !a->b_scale || ((uintptr_t)a->b_scale & 15) || (a->strideBScale & 3))) {
    set_error("alg_gemm_fp8: a_scale / b_scale are required (b_scale 16-byte aligned)");
    return ALG_EINVAL;
  }
  if (a->act < ALG_ACT_NONE || a->act > ALG_ACT_SILU) {
    set_error("alg_gemm_bf16: unknown activation %d", a->act);
    return ALG_EINVAL;
  }
  if ((a->flags & ALG_GEMM_PERMUTE_COLS) && (a->R || a->gate)) {
    set_error("alg_gemm_bf16: PERMUTE_COLS cannot be combined with residual/gate");
    return ALG_EINVAL;
  }
  if (a->R && a->act != ALG_ACT_NONE) {
    set_error("alg_gemm_bf16: an activation cannot be combined with the residual epilogue");
    return ALG_EINVAL;
  }
  if (a->gate && !a->R) {
    set_error("alg_gemm_bf16: gate needs a residual");
    return ALG_EINVAL;
  }
  // 8-byte epilogue accesses need 8-byte aligned quads whenever N is a multiple of 4
  if ((a->N & 3) == 0) {
    const bool bad = (a->ldc & 3) || (a->strideC & 3) || ((uintptr_t)a->C & 7) ||
                     (a->bias && !(a->flags & ALG_GEMM_BIAS_PER_ROW) && ((uintptr_t)a->bias & 7)) ||
                     (a->R && ((a->ldr & 3) || (a->strideR & 3) || ((uintptr_t)a->R & 7))) ||
                     (a->gate && ((a->strideGate & 3) ||
                                  ((uintptr_t)a->gate & ((a->flags & ALG_GEMM_GATE_F32) ? 15 : 7))));
    if (bad) {
      set_error("alg_gemm_bf16: C/bias/R/gate must be 8-byte aligned with ldc/ldr/strides multiples of 4 elements");
      return ALG_EINVAL;
    }
  }
  if (a->conv_wp) {
    const int cl = a->conv_cin_log2;
    const int taps = (cl >= 6 && cl <= 12) ? a->K >> cl : 0;
    const int kw = a->conv_kw;
    if (fp8 || (kw != 3 && kw != 4) || (taps != 3 * kw && taps != 9 * kw) || (taps << cl) != a->K ||
        (a->lda != ((int64_t)(kw - 2) << cl) && !(kw == 3 && a->lda == (2ll << cl))) || a->conv_wp < 3 ||
        a->conv_hpwp < 3 * a->conv_wp || a->act != ALG_ACT_NONE ||
        ((2ll * a->conv_hpwp + 2ll * a->conv_wp + 3) << cl) >= (1ll << 31)) {
      set_error("alg_gemm_bf16: bad convolution addressing (cin_log2=%d K=%d lda=%lld wp=%d hpwp=%d)", cl, a->K,
                (long long)a->lda, a->conv_wp, a->conv_hpwp);
      return ALG_EINVAL;
    }
  }
  if (validate_only) return ALG_OK;
  if ((int64_t)a->M * a->ldc >= (1ll << 31) || (a->R && (int64_t)a->M * a->ldr >= (1ll << 31))) {
    // Every argument check above has run on the WHOLE call, so a slab can only fail at launch (no error after
    // earlier slabs have already written C, which may alias R).  Per-row operands are A, C, R, a per-row bias, a_scale and
    // seg_split: each moves with the slab.  `gate` is per BATCH item (two segments), never per row: a caller that flattens
    // several samples into M shares one gate pair between them -- per-sample gates go through `batch` / strideGate.
    // The epilogue addresses C / R with 32-bit element offsets inside one batch item.  Taller operands (two or three CFG
    // samples of a 75,600- or 118,800-token sequence flattened into M) are cut along M into tile-aligned slabs, one launch
    // each on the same stream: rows are independent, every per-row operand just moves with the slab.
    const int64_t ld = (a->R && a->ldr > a->ldc) ? a->ldr : a->ldc;
    const int64_t rows = ((1ll << 31) - 1) / ld / BM * BM;
    if (a->conv_wp || rows <= 0) {
      set_error("alg_gemm_bf16: M*ldc must stay below 2^31 (32-bit epilogue offsets)");
      return ALG_ELIMIT;
    }
    const int64_t esz = fp8 ? 1 : 2;
    for (int64_t m0 = 0; m0 < a->M; m0 += rows) {
      alg_gemm_args c = *a;
      c.M = (int32_t)((a->M - m0) < rows ? (a->M - m0) : rows);
      c.A = (const char*)a->A + m0 * a->lda * esz;
      c.C = (char*)a->C + m0 * a->ldc * 2;
      if (a->R) c.R = (const char*)a->R + m0 * a->ldr * 2;
      if (a->bias && (a->flags & ALG_GEMM_BIAS_PER_ROW)) c.bias = (const char*)a->bias + m0 * 2;
      if (a->a_scale) c.a_scale = a->a_scale + m0;
      c.seg_split = a->seg_split > m0 ? (int32_t)(a->seg_split - m0) : 0;   // rows >= seg_split take gate[1]
      const int rc = gemm_entry(&c, stream, fp8);
      if (rc != ALG_OK) return rc;
    }
    return ALG_OK;
  }
  const int m_tiles = (a->M + BM - 1) / BM, n_tiles = (a->N + BN - 1) / BN;
  const int64_t nwg = (int64_t)m_tiles * n_tiles * a->batch;
  if (nwg > 0x7fffffff) {
    set_error("alg_gemm_bf16: grid too large");
    return ALG_ELIMIT;
  }
  hipStream_t s = (hipStream_t)stream;
  if (fp8) {   // schedule 9 (round 4: the asm loop on the block-scaled MFMA) needs two k-tiles of 128 and 32-bit byte offsets
    if (asm_pipe() && a->K >= 256 && 256 * a->lda + (int64_t)a->K < (1ll << 32) && 256 * a->ldb + (int64_t)a->K < (1ll << 32))
      return launch_gemm_p9_fp8(a, m_tiles, n_tiles, nwg, s);
    return launch_gemm_p6_fp8(a, m_tiles, n_tiles, nwg, s);
  }
  if (a->conv_wp) return launch_gemm_p6_conv(a, m_tiles, n_tiles, nwg, s);
  if (a->flags & ALG_GEMM_B_PACKED11) {
    // the caller packed B for schedule 11 (alg_pack_b_p11): that schedule or nothing (B cannot be read any other way)
    if (a->K < 128 || 256 * a->lda * 2 + (int64_t)a->K * 2 >= (1ll << 32) || a->strideB != 0 ||
        (a->flags & (ALG_GEMM_BIAS_PER_ROW | ALG_GEMM_PERMUTE_COLS)) || ((uintptr_t)a->B & 15)) {
      set_error("alg_gemm_bf16: a packed B (ALG_GEMM_B_PACKED11) needs K >= 128, 32-bit byte offsets inside a 256-row A panel, a B shared by "
                "the batch (strideB = 0) and a plain column layout (no per-row bias / column permutation)");
      return ALG_EINVAL;
    }
    return launch_gemm_p11(a, m_tiles, n_tiles, nwg, s);
  }
  switch (gemm_pipe()) {
    case 9:   // 4 waves, hand-written asm main loop; needs two k-tiles and 32-bit byte offsets inside a 256-row panel
    case 10:
      if (a->K >= 128 && 256 * a->lda * 2 + (int64_t)a->K * 2 < (1ll << 32) && 256 * a->ldb * 2 + (int64_t)a->K * 2 < (1ll << 32))
        return gemm_pipe() == 10 ? launch_gemm_p10(a, m_tiles, n_tiles, nwg, s) : launch_gemm_p9(a, m_tiles, n_tiles, nwg, s);
      return launch_gemm_p6(a, m_tiles, n_tiles, nwg, s);
    default: return launch_gemm_p6(a, m_tiles, n_tiles, nwg, s);  // 8-wave ping-pong over half-tiles
  }
}

extern "C" int alg_gemm_bf16(const alg_gemm_args* a, void* stream) { return gemm_entry(a, stream, false); }

// Two independent plain GEMMs in one persistent launch when schedule 9 can take both (no residual / activation / convolution,
// K >= 128, 32-bit offsets, no slab split); otherwise -- and always with ALG_GEMM_PIPE=6 -- simply one launch after the other.
// Either way every output element is computed exactly as alg_gemm_bf16 would compute it (bit-identical).
static bool pair_eligible(const alg_gemm_args* a) {
  return a && a->A && a->B && a->C && !a->R && !a->gate && a->act == ALG_ACT_NONE && !a->conv_wp && a->M > 0 && a->N > 0 &&
         a->batch > 0 && a->K >= 128 && a->K % 64 == 0 && 256 * a->lda * 2 + (int64_t)a->K * 2 < (1ll << 32) &&
         256 * a->ldb * 2 + (int64_t)a->K * 2 < (1ll << 32) && (int64_t)a->M * a->ldc < (1ll << 31) &&
         !(a->lda % 8 || a->ldb % 8 || a->strideA % 8 || a->strideB % 8 || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) &&
         ((a->N & 3) != 0 || !((a->ldc & 3) || (a->strideC & 3) || ((uintptr_t)a->C & 7) ||
                               (a->bias && !(a->flags & ALG_GEMM_BIAS_PER_ROW) && ((uintptr_t)a->bias & 7))));
}

extern "C" int alg_gemm_bf16_pair(const alg_gemm_args* a, const alg_gemm_args* b, void* stream) {
  // every argument check of alg_gemm_bf16 runs on BOTH calls before anything is launched
  int rc = gemm_entry(a, stream, false, true);
  if (rc == ALG_OK) rc = gemm_entry(b, stream, false, true);
  if (rc != ALG_OK) return rc;
  if (asm_pipe() && pair_eligible(a) && pair_eligible(b)) {
    const int64_t ta = (int64_t)((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN) * a->batch;
    const int64_t tb = (int64_t)((b->M + BM - 1) / BM) * ((b->N + BN - 1) / BN) * b->batch;
    if (ta + tb <= 0x7fffffff)
      return (gemm_pipe() == 10 ? launch_gemm_p10_pair : launch_gemm_p9_pair)(a, (a->M + BM - 1) / BM, (a->N + BN - 1) / BN, b, (b->M + BM - 1) / BM, (b->N + BN - 1) / BN,
                                 (hipStream_t)stream);
  }
  rc = gemm_entry(a, stream, false);
  return rc != ALG_OK ? rc : gemm_entry(b, stream, false);
}

extern "C" int alg_gemm_bf16_pair_qk(const alg_gemm_args* a, const alg_gemm_args* b, const alg_qk_norm_rope_args* e, void* stream) {
  int rc = gemm_entry(a, stream, false, true);
  if (rc == ALG_OK) rc = gemm_entry(b, stream, false, true);
  if (rc != ALG_OK) return rc;
  if (!e || !e->wq || !e->bq || !e->wk || !e->bk || e->heads <= 0 || e->text_len < 0 || e->text_len > a->M ||
      (e->cos_tab == nullptr) != (e->sin_tab == nullptr)) {
    set_error("alg_gemm_bf16_pair_qk: bad LayerNorm / rotary arguments");
    return ALG_EINVAL;
  }
  if (((uintptr_t)e->wq & 15) || ((uintptr_t)e->bq & 15) || ((uintptr_t)e->wk & 15) || ((uintptr_t)e->bk & 15) ||
      ((uintptr_t)e->cos_tab & 15) || ((uintptr_t)e->sin_tab & 15) || ((uintptr_t)a->C & 15)) {
    set_error("alg_gemm_bf16_pair_qk: pointers must be 16-byte aligned");
    return ALG_EINVAL;
  }
  if (a->N != 2 * e->heads * 64 || a->ldc != a->N || a->strideC != (int64_t)a->M * a->N || (a->flags & ALG_GEMM_PERMUTE_COLS) ||
      a->R || a->act != ALG_ACT_NONE) {
    set_error("alg_gemm_bf16_pair_qk: qk must be a plain GEMM onto the contiguous [batch][S][2][heads][64] tensor (N=%d heads=%d ldc=%lld)",
              a->N, e->heads, (long long)a->ldc);
    return ALG_EINVAL;
  }
  // the store loop owns whole head vectors and whole Q / K tiles: heads * 64 a multiple of the 256-column tile, staged epilogue
  // (N % 8 == 0 and 16-byte rows hold by the layout; a per-row bias would take the element-exact path)
  const bool fused = asm_pipe() && pair_eligible(a) && pair_eligible(b) && e->heads % 4 == 0 &&
                     !(a->bias && (a->flags & ALG_GEMM_BIAS_PER_ROW));
  if (fused) {
    const int64_t ta = (int64_t)((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN) * a->batch;
    const int64_t tb = (int64_t)((b->M + BM - 1) / BM) * ((b->N + BN - 1) / BN) * b->batch;
    if (ta + tb <= 0x7fffffff)
      return (gemm_pipe() == 10 ? launch_gemm_p10_pair_qk : launch_gemm_p9_pair_qk)(a, (a->M + BM - 1) / BM, (a->N + BN - 1) / BN, b, (b->M + BM - 1) / BM, (b->N + BN - 1) / BN, e,
                                    (hipStream_t)stream);
  }
  rc = alg_gemm_bf16_pair(a, b, stream);
  if (rc != ALG_OK) return rc;
  return alg_qk_norm_rope_scaled(a->C, e->wq, e->bq, e->wk, e->bk, e->cos_tab, e->sin_tab, a->batch, a->M, e->heads, e->text_len,
                                 e->eps, e->q_scale, stream);
}

extern "C" int alg_gemm_fp8(const alg_gemm_args* a, void* stream) { return gemm_entry(a, stream, true); }

// AutoencoderKLCogVideoX convolutions (CogVideoXCausalConv3d k = 3, upsampler Conv2d k = 3) as one GEMM launch over the
// padded grid: output row r = (y, x) of frame t reads input rows r + dt*Hp*Wp + dy*Wp + dx of frame t.
extern "C" int alg_conv_cl_bf16(const void* x, const void* w, const void* bias, const void* res, void* y, int frames,
                                int Hp, int Wp, int Cin, int Cout, int kt, int mode, void* stream) {
  const bool pair = mode == ALG_CONV_PAIR, down = mode == ALG_CONV_STRIDE2;
  if (frames <= 0 || Hp < 3 || Wp < 3 || (kt != 1 && kt != 3) || Cin < 64 || (Cin & (Cin - 1)) || Cout <= 0 || (Cout & 3) ||
      mode < 0 || mode > ALG_CONV_STRIDE2 || (pair && (((Hp * Wp) & 1) || Cout > 128)) ||
      (down && (kt != 1 || (Hp & 1) || (Wp & 1)))) {
    set_error("alg_conv_cl_bf16: bad shape frames=%d Hp=%d Wp=%d Cin=%d Cout=%d kt=%d (Cin a power of two >= 64, Cout %% 4 == 0)",
              frames, Hp, Wp, Cin, Cout, kt);
    return ALG_EINVAL;
  }
  alg_gemm_args a = {};
  a.A = x, a.B = w, a.C = y, a.bias = bias, a.R = res;
  const int vox = pair ? 2 : 1, kw = pair ? 4 : 3;  // voxels per GEMM row, taps along x
  a.lda = (int64_t)vox * Cin, a.ldb = (int64_t)kt * 3 * kw * Cin, a.ldc = (int64_t)vox * Cout, a.ldr = a.ldc;
  a.strideA = (int64_t)Hp * Wp * Cin, a.strideB = 0, a.strideC = (int64_t)Hp * Wp * Cout, a.strideR = a.strideC;
  a.M = Hp * Wp / vox, a.N = vox * Cout, a.K = kt * 3 * kw * Cin, a.batch = frames;
  a.act = ALG_ACT_NONE;
  a.conv_cin_log2 = __builtin_ctz((unsigned)Cin), a.conv_wp = Wp, a.conv_hpwp = Hp * Wp, a.conv_kw = kw;
  if (down) {
    // CogVideoXDownsample3D: pad (0, 1, 0, 1), Conv2d k3 s2 p0: output (Y, X) reads unpadded (2Y + dy, 2X + dx) = padded
    // rows (Wp + 1) + 2 (Y Wp + X) + dy Wp + dx: GEMM row m = Y Wp + X over the INPUT's pitch, two voxels apart
    a.A = (const bf16_t*)x + (int64_t)(Wp + 1) * Cin;
    a.lda = 2ll * Cin;
    a.M = (Hp - 2) / 2 * Wp;
    a.strideC = a.strideR = (int64_t)a.M * Cout;
  }
  return gemm_entry(&a, stream, false);
}
