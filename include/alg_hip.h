/*
 * alg_hip.h -- C ABI of libalg_hip.so, the MI355X (gfx950) native hot path of the ALG sampler.
 *
 * The reference (choi403/ALG) has no native/FFI layer of its own: its hot path is PyTorch ops called
 * from Python.  Each entry point below therefore cites the reference *call site* it replaces
 * (paths relative to the reference checkout).  Abbreviations:
 *   lp:  lp_utils.py            cog: pipeline_cogvideox_image2video_lowpass.py
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never allocates, frees or caches
 *     device memory: scratch space is a `workspace` argument sized by the matching alg_*_workspace_bytes()
 *     query, precomputed tables are a caller-owned blob filled by an explicit alg_*_build() call
 *   - enqueue-only on `stream` (a hipStream_t, may be NULL = default stream); no internal sync, no blocking
 *     copy: every entry point is legal inside a hipGraph stream capture
 *   - in/out may not alias unless stated
 *   - return 0 on success, negative ALG_E* on failure; message via alg_last_error() (thread local)
 *   - dtype codes: ALG_F32 = 0, ALG_BF16 = 1
 */
#ifndef ALG_HIP_H
#define ALG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALG_F32 0
#define ALG_BF16 1

#define ALG_OK 0
#define ALG_EINVAL (-1)   /* bad argument (shape, dtype, alignment, null pointer) */
#define ALG_ELAUNCH (-2)  /* HIP launch / runtime error */
#define ALG_ELIMIT (-3)   /* shape exceeds what the kernel supports (e.g. plane does not fit LDS) */

#define ALG_VERSION 110

int alg_version(void);
const char* alg_last_error(void);

/* Run-time options.  The library reads its ALG_* environment variables ONCE, when it is loaded; no launch path calls getenv.
 * A host that changes one of them afterwards calls alg_reload_env() (host-only, no GPU work; not to be called while another
 * thread is inside the library).  The default build knows seven, each selecting between bit-identical or documented-equivalent
 * schedules (README.md "Run-time options"): ALG_ATTN_SPLIT_TAIL, ALG_ATTN_PP (4 = the 8-wave pipelined statement on v_mfma_f32_32x32x16_bf16, 7 = the same construction on
 * v_mfma_f32_16x16x32_bf16 -- a call it declines (fewer than 12 KV tiles, 31-bit offsets, V^T pitch) runs the default, 4 --, 0 = the
 * straight loop), ALG_ATTN_VARIANT, ALG_ATTN128_PIPE,
 * ALG_ATTN128_Q64 (1 = the 64-queries-per-wave d = 128 kernel from 4,096 keys on, the default; 2 = for every call it can
 * take; 0 = off: the escape hatch back to the 32-query pipelined kernel), ALG_GEMM_PIPE, ALG_LOWPASS_PATH.  A value must be
 * a whole decimal integer the build knows; anything else (including "off", "1x", an empty string) leaves the default in
 * place. */
void alg_reload_env(void);

/* ------------------------------------------------------------------------------------------------
 * Low-pass filters on [planes, H, W] contiguous planes (a 4-D/5-D tensor viewed per (H, W) plane,
 * lp:31-37).  Planes resident in LDS: one workgroup per plane for small calls, a persistent grid of
 * register-blocked workgroups from 128 planes up (all bit-identical; planes beyond LDS size run
 * through global-memory passes).
 * ---------------------------------------------------------------------------------------------- */

/* lp:49-54  F.interpolate(bilinear, antialias) to (h1, w1) then back to (H, W).
 * h1, w1 are computed by the caller exactly as lp:51-52 (Python banker's rounding).
 * round_intermediate != 0 rounds the (h1, w1) intermediate to bf16 (the reference's two separate
 * interpolate calls each return a tensor of the input dtype); only meaningful for ALG_BF16. */
int alg_down_up(const void* in, void* out, int64_t planes, int H, int W, int h1, int w1, int dtype,
                int round_intermediate, const void* tables, void* workspace, int64_t workspace_bytes, void* stream);

/* The antialias tap tables of one (H, W) -> (h1, w1) -> (H, W) shape (lp:51-54: the weights F.interpolate(antialias=True)
 * derives per output index), as a blob the caller owns: alg_lowpass_tables_bytes() sizes it, alg_lowpass_tables_build()
 * fills it with one tiny kernel on `stream`, alg_down_up(tables=...) reads it on any later call of that shape (ordered
 * after the build on the stream, as usual).  tables = NULL is legal: alg_down_up then uses the kernels that derive the
 * taps per plane (identical bits, lower throughput on many-plane batches). */
int64_t alg_lowpass_tables_bytes(int H, int W, int h1, int w1);
int alg_lowpass_tables_build(void* tables, int64_t bytes, int H, int W, int h1, int w1, void* stream);

/* Scratch bytes alg_down_up needs for this call (0 for planes that fit LDS -- every latent-sized plane; pixel-sized planes
 * run through global-memory passes with fp32 intermediates in `workspace`). */
int64_t alg_down_up_workspace_bytes(int64_t planes, int H, int W, int h1, int w1);

/* lp:40-47  torchvision gaussian_blur(kernel_size=[k,k], sigma=[s,s]): reflect pad k/2, separable
 * correlation with g = exp(-0.5 (x/s)^2)/sum.  ksize must be odd (any size: more than 255 taps run through the global-memory
 * passes), ksize/2 < min(H, W), sigma > 0.  Weights are fp32 for every dtype (torchvision builds them in the image dtype:
 * for bf16 tensors see DESIGN.md section 2, "Rounding semantics"). */
int alg_gaussian_blur(const void* in, void* out, int64_t planes, int H, int W, int ksize, float sigma, int dtype,
                      void* workspace, int64_t workspace_bytes, void* stream);
int64_t alg_gaussian_blur_workspace_bytes(int64_t planes, int H, int W, int ksize);   /* 0 for LDS-sized planes */

/* ------------------------------------------------------------------------------------------------
 * cog:1091-1123  noise_pred.float(); chunk; CFG combine; CogVideoXDDIMScheduler.step (v-prediction,
 * eta = 0); cast back -- one fused elementwise pass, latents updated IN PLACE.
 *   n_pass = 3: u0 + g*(text - u)   (pred = [uncond_init, uncond, text], cog:1099-1102)
 *   n_pass = 2: u  + g*(text - u)   (cog:1096-1097)
 *   n_pass = 1: pred                (no CFG)
 *   x0 = sqrt_alpha_t * x - sqrt_beta_t * v ;  x <- coef_a * x + coef_b * x0
 * pred: [n_pass, numel] of pred_dtype;  latents: [numel] of lat_dtype.
 * ---------------------------------------------------------------------------------------------- */
int alg_cfg_ddim_step(const void* pred, int pred_dtype, void* latents, int lat_dtype, int n_pass, int64_t numel,
                      float guidance_scale, float sqrt_alpha_t, float sqrt_beta_t, float coef_a, float coef_b,
                      void* stream);

/* wan:919-924, hy:1254-1261  CFG combine in the prediction dtype (no .float()): out = u0 + g*(text - u), every
 * intermediate rounded to `dtype`.  pred [n_pass, numel] (n_pass 2 or 3), out [numel]. */
int alg_cfg_combine(const void* pred, void* out, int dtype, int n_pass, int64_t numel, float guidance_scale,
                    void* stream);

/* out = sum_i coefs[i] * xs[i]  (1..4 terms, each ALG_F32 or ALG_BF16).  Rounding follows torch eager: every product
 * is rounded to its tensor's dtype, the running sum is fp32, left to right, unfused; cast to out_dtype at the end.
 * Covers FlowMatchEulerDiscreteScheduler.step (hy:1265-1269: sample + (sigma_next - sigma) * v) and UniPC's
 * convert_model_output (wan:927: sample - sigma * v).
 * xs, coefs, dtypes are HOST arrays; the pointers inside xs are device pointers. */
int alg_lincomb(const void* const* xs, const float* coefs, const int* dtypes, int n_terms, void* out, int out_dtype,
                int64_t numel, void* stream);

/* wan:877-889, hy:1146-1160, 1230  CFG batch assembly in one launch (replaces cat([latents]*n) + cat(dim) + .to()):
 *   out[i, o, a, r] = a < A0 ? src0[i][o*s0_ostride + a*R + r] : src1[i][o*s1_ostride + (a1_off + a - A0)*R + r]
 * for i < n (<= 16), o < O, a < A0 + A1, r < R, cast to out_dtype.  src0 / src1 are HOST arrays of n device pointers
 * (one per output sample -- repeat a pointer to duplicate a sample across CFG passes).
 *   Wan    (channel concat [latents | condition]):      O=1, A0=16, A1=20, R=F*H*W
 *   Hunyuan (first-frame token replace [cond | lat 1:]): O=C, A0=cond frames, A1=F-1, R=H*W, a1_off=1 */
int alg_concat_cast(const void* const* src0, int dtype0, const void* const* src1, int dtype1, int n, int64_t O,
                    int64_t A0, int64_t A1, int64_t R, int64_t s0_ostride, int64_t s1_ostride, int64_t a1_off,
                    void* out, int out_dtype, void* stream);

/* wan:927  UniPCMultistepScheduler (bh1/bh2, predict_x0, solver_order <= 2) predictor or corrector update on fp32:
 *   out = (r*x - c*m0) - k * ( [m1] rho0*((m1 - m0)/rk)  +  [m_new] rho_new*(m_new - m0) )
 * m1 / m_new may be NULL (order-1 predictor: both NULL; order-1 corrector: m1 NULL). */
int alg_unipc_update(const float* x, const float* m0, const float* m1, const float* m_new, float* out,
                     int64_t numel, float r, float c, float k, float rk, float rho0, float rho_new, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Video-DiT forward building blocks (cog:1082-1090 self.transformer(...); the arithmetic is
 * diffusers' CogVideoXTransformer3DModel -- see DESIGN.md).  All activations bf16, fp32 accumulate.
 * ---------------------------------------------------------------------------------------------- */

#define ALG_ACT_NONE 0
#define ALG_ACT_GELU_TANH 1
#define ALG_ACT_SILU 2

#define ALG_GEMM_BIAS_PER_ROW 1   /* bias indexed by output row (used for the transposed V projection) */
#define ALG_GEMM_PERMUTE_COLS 4   /* store column n at n with bits 2 and 3 swapped (MFMA k-order for V^T) */
#define ALG_GEMM_GATE_SEG_STRIDE 16 /* gate[1] sits gate_seg_stride elements after gate[0] instead of N */
#define ALG_GEMM_B_PACKED11 32     /* B is NOT [N][K] but the panel alg_pack_b_p11 wrote from it (ldb ignored, strideB must be 0): the call runs
                                     GEMM schedule 11 -- 1 x 4 waves, the weight fetched straight into registers in MFMA-fragment order */
#define ALG_GEMM_GATE_F32 8       /* gate is float32 and C = bf16(R + gate * bf16(acc + bias)) with ONE final rounding
                                     (WanTransformerBlock: (x.float() + out * gate_msa).type_as(x)) */

typedef struct alg_gemm_args {
  const void* A;      /* [batch][M][K] bf16, row stride lda, batch stride strideA (elements) */
  const void* B;      /* [batch][N][K] bf16 (nn.Linear weight layout), ldb, strideB (0 = shared)  */
  void* C;            /* [batch][M][N] bf16, ldc, strideC */
  const void* bias;   /* [N] (or [M] with BIAS_PER_ROW) bf16, may be NULL */
  const void* R;      /* residual [batch][M][N] bf16 (ldr, strideR), may be NULL; may alias C   */
  const void* gate;   /* [batch][2][N] bf16: gate[0] for rows < seg_split, gate[1] otherwise; NULL = 1 */
  int64_t lda, ldb, ldc, ldr;
  int64_t strideA, strideB, strideC, strideR, strideGate;
  int32_t M, N, K, batch;
  int32_t seg_split;
  int32_t act;        /* ALG_ACT_* applied to (acc + bias) */
  int32_t flags;      /* ALG_GEMM_* */
  int64_t gate_seg_stride; /* with ALG_GEMM_GATE_SEG_STRIDE: elements between gate[0] and gate[1] (default N; 0 = one gate) */
  int32_t perm_col0;  /* with ALG_GEMM_PERMUTE_COLS: output column n is joint column perm_col0 + n of a wider permuted row
                         (C points at joint column 0); 0 for a stand-alone V^T */
  int32_t conv_cin_log2; /* implicit-GEMM convolution (alg_conv_bf16 sets these; 0 = plain GEMM): log2 of the channels per tap */
  const float* a_scale; /* alg_gemm_fp8 only: one scale per row of A, [batch][M] at batch stride strideAScale */
  const float* b_scale; /* alg_gemm_fp8 only: one scale per row of B, [batch][N] at batch stride strideBScale (0 = shared) */
  int64_t strideAScale, strideBScale;
  int32_t conv_wp;   /* rows of A between vertically adjacent taps (padded width) */
  int32_t conv_hpwp; /* rows of A between temporally adjacent taps (padded height * padded width) */
  int32_t conv_kw;   /* taps along x: 3, or 4 when one A row holds two neighbouring voxels (lda = 2 * channels) */
  int32_t reserved1;
} alg_gemm_args;

/* C = R + gate * act(A @ B^T + bias)   (bias optional; either act or the residual(+gate) form).  K % 64 == 0, lda/ldb % 8 == 0,
 * A and B 16-byte aligned.  M and N are arbitrary (edge tiles clamp loads and guard stores). */
int alg_gemm_bf16(const alg_gemm_args* args, void* stream);

/* Packs an nn.Linear weight W [N][K] (bf16, row pitch ldb elements) for GEMM schedule 11 (ALG_GEMM_B_PACKED11): per (256-row tile of W,
 * 64-deep k-tile) 32 KiB in MFMA-fragment order, rows past N zero.  alg_pack_b_p11_bytes(N, K) = ceil(N / 256) * (K / 64) * 32768 is the
 * size of `packed` (caller-owned, 16-byte aligned).  K % 64 == 0, K >= 128.  Done once when a model is loaded; the reference's
 * nn.Linear weights (e.g. the attn1.to_q / ff.net.* tensors behind /root/reference/pipeline_cogvideox_image2video_lowpass.py:1082) keep
 * their values -- only the order in memory changes. */
int64_t alg_pack_b_p11_bytes(int N, int K);
int alg_pack_b_p11(const void* w, void* packed, int N, int K, int64_t ldb, void* stream);

/* Two independent alg_gemm_bf16 calls as ONE persistent launch when both are plain (no residual, gate, activation or
 * convolution addressing; the Q|K and V^T projections of a DiT block, cog:1082-1090, read the same activations): the tiles of
 * `b` follow the tiles of `a` in the tile list, so the two launches' partial last rounds become one.  Every output element is
 * computed exactly as by the two separate calls (bit-identical); calls the pair form cannot take run one after the other.
 * C of one problem must not alias an operand of the other. */
int alg_gemm_bf16_pair(const alg_gemm_args* a, const alg_gemm_args* b, void* stream);

/* alg_gemm_bf16_pair with the CogVideoX block's per-head QK LayerNorm + rotary embedding (alg_qk_norm_rope_scaled; diffusers
 * CogVideoXAttnProcessor2_0: norm_q / norm_k / apply_rotary_emb, cog:1082-1090) applied to problem `qk` inside its store loop:
 * qk->C is the [batch][S][2][heads][64] tensor (M = S, N = ldc = 2*heads*64, strideC = S*N), written ONCE in its final
 * form.  Bit-identical to alg_gemm_bf16_pair followed by alg_qk_norm_rope_scaled(qk->C, ..., qk->batch, qk->M, heads,
 * text_len, eps, q_scale) -- which is exactly what runs when the fused form cannot take the call (heads % 4 != 0, schedule 6,
 * a pair the persistent launch cannot take). */
typedef struct alg_qk_norm_rope_args {
  const void *wq, *bq, *wk, *bk;      /* [64] bf16 LayerNorm weight / bias of norm_q and norm_k */
  const float *cos_tab, *sin_tab;      /* [S - text_len][64] float32, or both NULL (no rotary embedding) */
  int32_t heads, text_len;
  float eps, q_scale;                  /* q_scale: see alg_qk_norm_rope_scaled (1 = plain) */
} alg_qk_norm_rope_args;
int alg_gemm_bf16_pair_qk(const alg_gemm_args* qk, const alg_gemm_args* vt, const alg_qk_norm_rope_args* e, void* stream);

/* BASELINE config 5 (fp8 weights on the CDNA4 fp8 MFMA): same contract and epilogues with A and B holding OCP e4m3 bytes
 * (lda / ldb / strides in elements = bytes) and per-row float32 scales: C = epilogue(a_scale[m] * b_scale[n] * (A @ B^T)).
 * K % 128 == 0.  Operands come from alg_quantize_fp8_rows (activations: per token, weights: per output channel). */
int alg_gemm_fp8(const alg_gemm_args* args, void* stream);

/* Row-wise dynamic quantisation to OCP e4m3: scale[r] = max|x[r]| / 448 (1 if the row is zero), q = e4m3(x / scale).
 * x: rows of K bf16 at stride x_rstride (elements); q: rows of K bytes, contiguous; scale: [rows] float32. */
int alg_quantize_fp8_rows(const void* x, int64_t x_rstride, void* q, float* scale, int64_t rows, int K, void* stream);

/* Full (unmasked) softmax attention, head_dim 64.
 *   q, k : bf16, element (b, s, h, d) at  base + b*q_bstride + s*q_rstride + h*64 + d   (same strides for k)
 *   vt   : bf16 V transposed, element (b, h, d, s') at base + b*vt_bstride + (h*64+d)*vt_rstride + perm(s'),
 *          perm = swap of index bits 2 and 3 (what alg_gemm_bf16 + ALG_GEMM_PERMUTE_COLS writes);
 *          columns >= S up to the next multiple of 64 must be readable and finite (zero)
 *   o    : bf16, element (b, s, h, d) at base + b*o_bstride + s*o_rstride + h*64 + d
 * softmax(q k^T * scale) v, fp32 accumulate, P rounded to bf16 before P@V. */
int alg_flash_attn_d64(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S,
                       int64_t q_bstride, int64_t q_rstride, int64_t vt_bstride, int64_t vt_rstride,
                       int64_t o_bstride, int64_t o_rstride, float scale, void* stream);

/* flags = ALG_ATTN_Q_PRESCALED: q already carries scale * log2(e) (alg_qk_norm_rope_scaled); `scale` is ignored and the
 * scores come out of the MFMA in log2 units.  flags = 0 is alg_flash_attn_d64.
 * workspace: when the (head, query-block) units of a launch leave the last round of workgroups nearly empty (C2: 840 units
 * per XCD on 64 slots), the last units are cut along KV into a second launch whose fp32 partials live in `workspace`
 * (alg_flash_attn_d64_workspace_bytes; 0 = this shape is one launch).  workspace = NULL (what alg_flash_attn_d64 passes)
 * runs everything as one launch: same result up to the fp32 summation order of those last units. */
#define ALG_ATTN_Q_PRESCALED 1
int alg_flash_attn_d64_ex(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int S,
                          int64_t q_bstride, int64_t q_rstride, int64_t vt_bstride, int64_t vt_rstride,
                          int64_t o_bstride, int64_t o_rstride, float scale, int flags, void* workspace,
                          int64_t workspace_bytes, void* stream);
int64_t alg_flash_attn_d64_workspace_bytes(int batch, int heads, int S, int flags);

/* wan:910-917 (WanTransformer3DModel self- and cross-attention, head_dim 128; diffusers WanAttnProcessor SDPA)
 * Same contract as alg_flash_attn_d64 with head_dim 128 and separate query / key lengths:
 *   q : element (b, s, h, d) at q + b*q_bstride + s*q_rstride + h*128 + d,  s < Sq
 *   k : likewise with k_bstride / k_rstride, s < Skv
 *   vt: V transposed, element (b, h, d, s) at vt + b*vt_bstride + (h*128 + d)*vt_rstride + perm(s), perm swaps index
 *       bits 2 and 3 (what alg_gemm_bf16 writes with ALG_GEMM_PERMUTE_COLS); vt_rstride >= Skv rounded up to 64 and the
 *       padding columns must hold finite values (they are multiplied by p = 0)
 *   o : element (b, s, h, d) at o + b*o_bstride + s*o_rstride + h*128 + d */
int alg_flash_attn_d128(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                        int64_t q_bstride, int64_t q_rstride, int64_t k_bstride, int64_t k_rstride, int64_t vt_bstride,
                        int64_t vt_rstride, int64_t o_bstride, int64_t o_rstride, float scale, void* stream);

/* The same kernel with grouped-query attention (kv_group query heads share one K / V head; k and vt then hold heads / kv_group
 * heads) and an optional causal mask (query i sees keys 0..i; needs Sq == Skv): the Llama-3 tower of HunyuanVideo's prompt
 * encoder (reference pipeline_hunyuan_video_image2video_lowpass.py:282-420, transformers LlavaForConditionalGeneration). */
int alg_flash_attn_d128_ex(const void* q, const void* k, const void* vt, void* o, int batch, int heads, int Sq, int Skv,
                           int64_t q_bstride, int64_t q_rstride, int64_t k_bstride, int64_t k_rstride, int64_t vt_bstride,
                           int64_t vt_rstride, int64_t o_bstride, int64_t o_rstride, float scale, int kv_group, int causal,
                           void* stream);

/* Two key / value sets for the same queries, the two attention outputs added: the image + text cross-attention of the Wan I2V DiT
 * (wan:910-917 calls WanTransformer3DModel; its attention processor computes sdpa(q, k_img, v_img) + sdpa(q, k, v) on bf16 tensors)
 * as ONE launch -- o = bf16(bf16(attn(q, k, vt)) + bf16(attn(q, k2, vt2))), bit for bit what two alg_flash_attn_d128 calls and
 * alg_lincomb give.  Layouts as alg_flash_attn_d128; both sets are meant to be short (encoder tokens). */
int alg_flash_attn_d128_dual(const void* q, const void* k, const void* vt, int Skv, int64_t k_bstride, int64_t k_rstride,
                             int64_t vt_bstride, int64_t vt_rstride, const void* k2, const void* vt2, int Skv2, int64_t k2_bstride,
                             int64_t k2_rstride, int64_t vt2_bstride, int64_t vt2_rstride, void* o, int batch, int heads, int Sq,
                             int64_t q_bstride, int64_t q_rstride, int64_t o_bstride, int64_t o_rstride, float scale, void* stream);

/* Llama rotary embedding ("rotate_half" form) in place on x [rows][heads][128] bf16 (row stride x_rstride elements):
 * x' = x * cos[pos[r]] + rotate_half(x) * sin[pos[r]], every product and the sum rounded to bf16 like the eager graph;
 * cos / sin: fp32 tables [positions][128]; pos: int32 [rows]. */
int alg_rope_half(void* x, const float* cos_tab, const float* sin_tab, const int* pos, int64_t rows, int heads,
                  int64_t x_rstride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Text encoders of the conditioning front-end (transformers T5EncoderModel, cog:228-268 `_get_t5_prompt_embeds`;
 * UMT5EncoderModel, wan:185-234 `_get_t5_prompt_embeds`): the small kernels around alg_gemm_bf16
 * ------------------------------------------------------------------------------------------------ */

/* out[r][:] = table[ids[r]][:]  (nn.Embedding; ids int64, clamped into [0, vocab)), rows of D bf16, D % 8 == 0. */
int alg_embed_rows(const int64_t* ids, const void* table, void* out, int64_t n, int D, int vocab, void* stream);

/* T5LayerNorm: y = bf16( bf16(x * rsqrt(mean(x^2) + eps)) * weight ), variance in fp32, no mean subtraction, no bias. */
int alg_t5_layernorm(const void* x, const void* weight, void* y, int64_t rows, int D, float eps, void* stream);

/* T5Attention / CLIPAttention (eager graph; head_dim 64 with L <= 512, or 80 with L <= 448): for batch b, head h
 *   s = bf16(q k^T) [* scale -> bf16]  + bias_table[rel_bucket[j - i + L - 1]][h] -> bf16;  masked keys (key_mask[b][j]
 *   == 0) get probability 0, and with `causal` keys j > i too (CLIPTextModel);  p = bf16(softmax_fp32(s));  out = bf16(p v)
 * q / k / v: element (b, i, h, d) at ptr + (b*L + i)*qkv_rstride + h*head_dim + d (three pointers into one fused QKV buffer);
 * out likewise with out_rstride; bias_table: [buckets][heads] bf16 (relative_attention_bias.weight) or NULL;
 * rel_bucket: int32 [2L - 1], the bucket of relative position (key - query); key_mask: int32 [batch][L] or NULL. */
int alg_attn_bias(const void* q, const void* k, const void* v, void* out, const void* bias_table, const int* rel_bucket,
                  const int* key_mask, int batch, int heads, int head_dim, int L, int64_t qkv_rstride, int64_t out_rstride,
                  float scale, int causal, void* stream);

/* CLIPTextModel's `quick_gelu` (hy:421-452 `text_encoder_2`): x = bf16(x * bf16(sigmoid(bf16(1.702 x)))), in place. */
int alg_quick_gelu(void* x, int64_t numel, void* stream);

/* out = bf16(a * b) (T5DenseGatedActDense: gelu_new(wi_0 x) * wi_1 x). */
int alg_mul_bf16(const void* a, const void* b, void* out, int64_t numel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * AutoencoderKLCogVideoX decoder (diffusers; call site cog:428-433 `frames = self.vae.decode(latents).sample`)
 * Activations are channels-last bf16 over a padded grid (Hp, Wp) = (H + 2, W + 2):
 *   padded  [T + 2][Hp][Wp][C]  zero borders, frames 0 and 1 repeat frame 0 (causal padding)  -- convolution input
 *   virtual [T][Hp][Wp][C]      rows with y >= H or x >= W are don't-care                      -- convolution output
 * ------------------------------------------------------------------------------------------------ */

/* CogVideoXCausalConv3d (kt = 3: 3x3x3, first frame repeated twice in front, zero spatial padding) and the upsamplers'
 * Conv2d 3x3 (kt = 1), as one implicit-GEMM launch on the MFMA GEMM kernel:
 *   x : padded input [frames + kt - 1][Hp][Wp][Cin], readable for 2*Wp + 2 rows past its end
 *   w : [Cout][kt*9][Cin] bf16 (tap-major (dt, dy, dx), channels innermost);  bias: [Cout] or NULL
 *   y : virtual output [frames][Hp][Wp][Cout];  res: optional residual in y's layout, y = res + conv (may alias y)
 * Cin a power of two >= 64 (pad thinner inputs with zero channels), Cout % 4 == 0.
 * mode ALG_CONV_PAIR (Cout <= 128; the GEMM tile is 256 columns wide): one GEMM row produces TWO neighbouring voxels, so a
 * 128-channel convolution fills the tile: w is then [2*Cout][kt*3*4][Cin] with rows [0, Cout) = the kernel at dx 0..2
 * (dx 3 zero) and rows [Cout, 2*Cout) = the kernel at dx 1..3 (dx 0 zero), bias is [2*Cout] (the bias twice), Hp*Wp must
 * be even, and x must be readable for 2*Wp + 3 rows past its end.  Same results, 4/3 of the useful MFMA work instead of 2x.
 * mode ALG_CONV_STRIDE2 (kt = 1; CogVideoXDownsample3D: pad (0,1,0,1) + Conv2d k3 s2 p0): x is the padded input of the
 * [H][W] activation (H, W even), y has (H/2)*Wp rows per frame -- output (Y, X) at row Y*Wp + X, i.e. at the INPUT's pitch
 * (alg_vae_repitch moves it to the standard layout of the half-resolution level). */
#define ALG_CONV_PLAIN 0
#define ALG_CONV_PAIR 1
#define ALG_CONV_STRIDE2 2
int alg_conv_cl_bf16(const void* x, const void* w, const void* bias, const void* res, void* y, int frames, int Hp,
                     int Wp, int Cin, int Cout, int kt, int mode, void* stream);

typedef struct alg_vae_geom {
  int32_t frames, H, W, C;     /* activation extent; C a power of two in [128, 2048] (32 groups) */
  int32_t first_len, seg_len;  /* GroupNorm segments in frames: [0, first_len), then seg_len each -- the published decoder
                                  normalises each batch of latent frames (3 first, then 2) on its own */
  int32_t lat_first_single;    /* frame t came from latent frame (t ? 1 + (t-1)/lat_rate : 0) if set, else t/lat_rate */
  int32_t lat_rate, lat_scale; /* temporal and spatial upsampling factors relative to the latent */
  int32_t lat_h, lat_w;        /* latent grid (the conditioning tensor is padded: [L + 2][lat_h + 2][lat_w + 2][2C]) */
} alg_vae_geom;

/* GroupNorm(32 groups) statistics of a virtual-layout activation per (segment, group): stats[seg][32][2] = (mean, rstd).
 * Deterministic (fixed-order partial sums, final reduction in double).  workspace: alg_vae_groupnorm_workspace() bytes. */
int64_t alg_vae_groupnorm_workspace(const alg_vae_geom* g);
int alg_vae_groupnorm_stats(const void* x, const alg_vae_geom* g, float eps, void* workspace, float* stats, void* stream);

/* CogVideoXSpatialNorm3D (+ SiLU): out = act(GroupNorm(x) * conv_y(zq) + conv_b(zq)), virtual -> padded.  zyb holds
 * [conv_y(zq) | conv_b(zq)] (2C channels) at latent resolution in the padded latent layout; each tensor op rounds to bf16
 * as the eager reference does. */
int alg_vae_spatial_norm(const void* x, const float* stats, const void* gamma, const void* beta, const void* zyb,
                         void* out, const alg_vae_geom* g, int silu, void* stream);

/* The encoder's plain GroupNorm (+ SiLU), virtual -> padded: same kernel without the conditioning (lat_* ignored). */
int alg_vae_group_norm(const void* x, const float* stats, const void* gamma, const void* beta, void* out,
                       const alg_vae_geom* g, int silu, void* stream);

/* virtual -> padded copy (zero borders, causal frames), for convolutions that read an un-normalised activation. */
int alg_vae_pad(const void* x, void* out, int frames, int H, int W, int C, void* stream);

/* Rows written at another pitch (ALG_CONV_STRIDE2: src_rows rows per frame, src_wp per line) -> the standard virtual
 * layout [frames][H + 2][W + 2][C]. */
int alg_vae_repitch(const void* x, void* out, int frames, int H, int W, int C, int src_rows, int src_wp, void* stream);

/* virtual [frames][H+2][W+2][C] -> planes [C][frames][H][W] bf16 (the encoder's moments, `AutoencoderKLCogVideoX.encode`). */
int alg_vae_unpack_planes(const void* x, void* out, int frames, int H, int W, int C, void* stream);

/* CogVideoXUpsample3D nearest-neighbour part: virtual [T][H+2][W+2][C] -> padded [frames_out][2H+2][2W+2][C] (no time
 * padding, the convolution that follows is 2-D); compress_time doubles every frame but the first (first_single) . */
int alg_vae_upsample(const void* x, void* out, int frames_out, int H, int W, int C, int compress_time, int first_single,
                     void* stream);

/* cog:429-430: z element (c, l, y, x) at z + c*c_stride + l*frame_stride + y*w + x, times `scale` (1 / scaling_factor,
 * one bf16 rounding) -> padded [frames + 2][h + 2][w + 2][64] with channels >= `channels` zero. */
int alg_vae_pack_latent(const void* z, int64_t c_stride, int64_t frame_stride, void* out, int frames, int h, int w,
                        int channels, float scale, void* stream);

/* virtual [frames][H+2][W+2][4] (RGB + pad channel) -> bf16 [3][frames][H][W] (decode().sample) or, with to_uint8, the
 * writer's uint8 [frames][H][W][3] (VideoProcessor.postprocess_video + run:121-125). */
int alg_vae_unpack_video(const void* x, void* out, int frames, int H, int W, int to_uint8, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HunyuanVideo DiT building blocks (diffusers HunyuanVideoTransformer3DModel; call site hy:1243-1252)
 * ------------------------------------------------------------------------------------------------ */

/* HunyuanVideoAttnProcessor2_0 qk norm (RMSNorm over each 128-wide head, weight [128]) + rotary embedding, in place:
 * row r of batch b = heads*128 bf16 at x + b*x_bstride + r*x_rstride; rope = x*cos + rot(x)*sin on interleaved pairs
 * with fp32 tables [rope_tokens][128], applied to rows r < rope_tokens only (latent tokens; text tokens follow them). */
int alg_headnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab, int64_t x_rstride,
                      int64_t x_bstride, int batch, int rows, int heads, int rope_tokens, float eps, void* stream);

/* HunyuanVideoTokenRefiner pooled prompt: out[b][d] = mean over the first valid[b] tokens of x[b][.][d] (bf16). */
int alg_masked_mean(const void* x, const int* valid, void* out, int batch, int L, int D, void* stream);

/* y = silu(x) on bf16 (AdaLayerNormZero: linear(silu(emb))). */
int alg_silu(const void* x, void* y, int64_t numel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Wan 2.1 DiT building blocks (diffusers WanTransformer3DModel; call site wan:910-917)
 * ------------------------------------------------------------------------------------------------ */

/* WanTransformerBlock norm1/norm2/norm3, norm_out:  y = bf16( FP32LayerNorm(x) [* weight + bias] [* (1 + scale[b]) +
 * shift[b]] ), the whole chain in fp32.  x, y: [batch][rows][D] bf16 contiguous; weight/bias: [D] float32 or NULL;
 * scale/shift: float32 vectors of batch b at scale + b*mod_bstride (NULL = no modulation).  Register-resident rows when
 * D % 512 == 0 (D/512 in {1,2,3,4,6,8,10,12}), a strided three-pass kernel otherwise. */
int alg_layernorm_mod_f32(const void* x, void* y, const float* weight, const float* bias, const float* scale,
                          const float* shift, int64_t mod_bstride, int batch, int rows, int D, float eps, void* stream);

/* The same norm feeding an fp8 GEMM: instead of the bf16 tensor, writes what alg_quantize_fp8_rows would make of it
 * (q8: [batch*rows][D] OCP e4m3 bytes, q8_scale: [batch*rows] float32 = amax / 448 of the bf16-rounded row), bit for bit,
 * without the bf16 round trip through HBM.  D % 512 == 0 only. */
int alg_layernorm_mod_f32_fp8(const void* x, void* q8, float* q8_scale, const float* weight, const float* bias,
                              const float* scale, const float* shift, int64_t mod_bstride, int batch, int rows, int D,
                              float eps, void* stream);

/* WanAttnProcessor norm_q / norm_k (RMSNorm across all heads) + rotary embedding, in place:
 *   x = rope( bf16( bf16(x * rsqrt(mean(x^2) + eps)) * weight ) ),  x: [batch*rows] rows of D bf16 at stride x_rstride;
 * rope multiplies the interleaved pairs (2j, 2j+1) of every 128-wide head by cos/sin[token][j] (fp32 tables [rows][64],
 * token = row % rows); cos_tab NULL = no rope (cross-attention q, text / image k). */
int alg_rmsnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab, int64_t x_rstride,
                     int batch, int rows, int D, float eps, void* stream);

/* out[l][b][j][d] = table[l][j][d] + float(vec[b][(vec_per_j ? j*D : 0) + d]):  (scale_shift_table + temb.float()) of
 * every block in one launch (J = 6, vec = timestep_proj), and of the output head (J = 2, vec = temb). */
int alg_wan_modulation(const float* table, const void* vec, float* out, int layers, int batch, int J, int D,
                       int vec_per_j, void* stream);

/* Conv3d(kernel = stride = (1, ph, pw)) patch gather: out[n][(f, gy, gx)][c*ph*pw + py*pw + px] = in[n][c][f][..][..],
 * row length Kpad >= C*ph*pw (zero padded so the patch-embed GEMM sees K % 64 == 0). */
int alg_patchify3d(const void* in, void* out, int n, int C, int F, int H, int W, int ph, int pw, int Kpad, void* stream);

/* proj_out rows (row stride ldin) -> [n][C][F][H][W] bf16; the row holds (py*pw + px)*C + c (Wan, channel_major = 0) or
 * c*ph*pw + py*pw + px (HunyuanVideo, channel_major = 1). */
int alg_unpatchify3d(const void* in, int64_t ldin, void* out, int n, int C, int F, int H, int W, int ph, int pw,
                     int channel_major, void* stream);

/* Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) in float32: out[n][dim] = [cos | sin]. */
int alg_timestep_embedding_f32(const float* t, float* out, int n, int dim, void* stream);

/* float32 linear for the modules diffusers keeps in fp32 (time_embedder): y = act(x W^T + b), x [M][K], W [N][K];
 * act 0 none, 1 SiLU.  Any of y (fp32), y_bf16 (= bf16(y)), y_silu_bf16 (= bf16(silu(bf16(y)))) may be NULL. */
int alg_linear_f32(const float* x, const float* W, const float* b, float* y, void* y_bf16, void* y_silu_bf16, int M,
                   int N, int K, int act, void* stream);

/* exact (erf) GELU in place on bf16 (WanImageEmbedding feed-forward). */
int alg_gelu_erf(void* x, int64_t numel, void* stream);

/* y = LayerNorm(x; weight, bias, eps) * (1 + scale[seg]) + shift[seg]       (CogVideoXLayerNormZero / AdaLayerNorm)
 * x, y: [batch][rows][D] bf16, rows contiguous, batch strides x_bstride / y_bstride (elements);
 * weight/bias: [D] bf16 (may be NULL);
 * scale/shift: [batch][2][D] bf16 at batch stride mod_bstride (seg 0 = rows < seg_split, seg 1 = the rest),
 * NULL = plain LayerNorm.  D % 512 == 0, D <= 8192. */
int alg_layernorm_modulate(const void* x, void* y, const void* weight, const void* bias, const void* scale,
                           const void* shift, int64_t mod_bstride, int batch, int rows, int D, int64_t x_bstride,
                           int64_t y_bstride, int seg_split, float eps, void* stream);
/* same with an explicit distance (elements) between the two segments' vectors (seg_stride = D above; 0 = one vector for
 * all rows).  HunyuanVideo token_replace: rows < seg_split (first-frame tokens) take the timestep-0 modulation. */
int alg_layernorm_modulate_seg(const void* x, void* y, const void* weight, const void* bias, const void* scale,
                               const void* shift, int64_t mod_bstride, int64_t seg_stride, int batch, int rows, int D,
                               int64_t x_bstride, int64_t y_bstride, int seg_split, float eps, void* stream);

/* In place on qk: [batch][S][2][heads][64] bf16 (q then k per token):
 * per-head LayerNorm(64) with (wq,bq) / (wk,bk), then RoPE (cos/sin fp32 [S - text_len][64], interleaved-pair
 * convention) on tokens >= text_len. */
int alg_qk_norm_rope(void* qk, const void* wq, const void* bq, const void* wk, const void* bk, const float* cos_tab,
                     const float* sin_tab, int batch, int S, int heads, int text_len, float eps, void* stream);

/* The same with Q multiplied by q_scale inside its last rounding (K untouched): q_scale = softmax_scale * log2(e) lets
 * alg_flash_attn_d64_ex(ALG_ATTN_Q_PRESCALED) form its probabilities as exp2(score) with no per-score multiply. */
int alg_qk_norm_rope_scaled(void* qk, const void* wq, const void* bq, const void* wk, const void* bk, const float* cos_tab,
                            const float* sin_tab, int batch, int S, int heads, int text_len, float eps, float q_scale,
                            void* stream);

/* Patch gather for the patch-embed GEMM (cog:1060-1070 batch assembly folded in, no materialised cat):
 * out[n][(f, gy, gx)][c*p*p + py*p + px]; channels [0, C) come from latents (sample stride lat_bstride, 0 =
 * broadcast), channels [C, 2C) from cond[n] (cond_ptrs: n device pointers to [F][C][H][W] tensors). */
int alg_patchify(const void* latents, int64_t lat_bstride, const void* const* cond_ptrs, void* out, int n_samples,
                 int frames, int C, int H, int W, int p, void* stream);

/* proj_out rows [n][(f,gy,gx)][c*p*p + py*p + px] -> [n][F][C][H][W] bf16. */
int alg_unpatchify(const void* in, void* out, int n_samples, int frames, int C, int H, int W, int p, void* stream);

/* CogVideoX 1.5 (`patch_size_t`; CogVideoXPatchEmbed's Linear over (c, t, py, px) and the matching unpatchify): frames
 * are folded in groups of p_t: out[n][(f / p_t, gy, gx)][((c*p_t + f % p_t)*p + py)*p + px];  p_t = 1 is the above. */
int alg_patchify_t(const void* latents, int64_t lat_bstride, const void* const* cond_ptrs, void* out, int n_samples,
                   int frames, int C, int H, int W, int p, int p_t, void* stream);
int alg_unpatchify_t(const void* in, void* out, int n_samples, int frames, int C, int H, int W, int p, int p_t,
                     void* stream);

/* Timesteps(dim, flip_sin_to_cos, freq_shift=0): out [n][dim] bf16 sinusoid of t[n] (fp32 timesteps). */
int alg_timestep_embedding(const float* t, void* out, int n, int dim, int flip_sin_to_cos, void* stream);

/* ---- Wan 2.1 VAE (diffusers AutoencoderKLWan; reference pipeline_wan_image2video_lowpass.py:426-430 encode, :959 decode) ----
 * Its convolutions are alg_conv_cl_bf16 / alg_gemm_bf16 launches; these two are the kernels specific to it. */

/* WanRMS_norm (+ the SiLU that follows it in every residual block): y[r][c] = act(x[r][c] * sqrt(C) / max(||x[r][:C]||_2,
 * 1e-12) * gamma[c]) for c < C and 0 for the padding channels C <= c < Cp.  x, y: [rows][Cp] bf16, gamma: [Cp] bf16. */
int alg_rms_norm_rows(const void* x, const void* gamma, void* y, int64_t rows, int C, int Cp, int silu, void* stream);

/* Row softmax of the VAE mid-block attention over scores split in two bf16 matrices (hi = -neg_hi, lo): p[r][j] =
 * softmax_j((lo - neg_hi) * scale), zero in the padding columns cols <= j < ld.  All three [rows][ld] bf16. */
int alg_softmax_hilo(const void* neg_hi, const void* lo, void* p, int64_t rows, int cols, int64_t ld, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Box calibration for the benchmark line (no reference call site: the reference times nothing, readme.md:1-170; SURVEY 8d asks
 * for rooflines measured on the box).  csrc/calibrate.hip.
 * ---------------------------------------------------------------------------------------------- */

/* One launch of a register-only v_mfma_f32_32x32x16_bf16 loop on pseudo-random bf16 operands: `blocks` workgroups of 256
 * threads (0 = two per CU, i.e. two waves per SIMD); every wave issues 32 * iters MFMAs = 32 * iters * 32768 FLOP.  The caller
 * brackets the launch with events.  clocks (optional): uint64 [blocks][4] = {shader cycles, constant-rate ticks} at the start and
 * at the end of each workgroup.  Returns the number of workgroups launched (> 0) or a negative ALG_E* code. */
int alg_calib_mfma_bf16(float* sink, int iters, unsigned seed, int blocks, uint64_t* clocks, void* stream);

/* Rate of the constant counter of the clock taps in kHz (100,000 on MI355X); 0 when unknown. */
int alg_wall_clock_khz(void);

/* While `buffer` is non-NULL, every launch of the pipelined d = 64 attention and of the 64-query d = 128 attention has one lane
 * of each workgroup whose index is a multiple of 64 -- the first `slots` of them: one workgroup owns a slot -- store {shader
 * cycles, constant-rate ticks} at its start and end into buffer[block / 64][4] (uint64): d cycles / d ticks = the shader clock
 * that kernel ran at.  Results are unaffected.  A launch on a CAPTURING stream never takes the tap (the pointer would be baked
 * into the graph and outlive the buffer).  Host-only call; NULL (the default) switches the taps off; the caller keeps the
 * buffer alive until it has done so. */
void alg_attn_clock_tap(uint64_t* buffer, int slots);

#ifdef __cplusplus
}
#endif
#endif /* ALG_HIP_H */
