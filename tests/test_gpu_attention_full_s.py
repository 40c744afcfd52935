"""d = 128 flash attention against fp32 SDPA at the FULL key counts of BASELINE configs C3 / C5 / C4 (VERDICT r2 weak 4 /
next 2c): 32,760 (Wan 480p), 75,600 (Wan 720p) and 118,800 + 256 (HunyuanVideo 720p latent + prompt tokens) keys, for the
default kernel (`attention128.hip`) AND the 64-query kernel (`attention128_q64.hip`, ALG_ATTN128_Q64=1), plus the grouped-
query causal form the Llava-Llama-3 text encoder launches.

The reference is torch's own fp32 matmul / softmax on the GPU over EVERY row of the sampled heads (the scores of a query
chunk are a [chunk, S] fp32 matrix: 2048 x 118,800 x 4 B = 0.97 GB), so a few dozen corrupted tokens among 10^5 -- the
round-2 flake's signature -- fail a PER-ROW bound; the global norm is not what is asserted.  Queries are scaled so that
softmax rows are peaked (an effective support of a few hundred keys): outputs are O(0.1), not the O(S^-1/2) mean of a
near-uniform row, and the lazy running max takes its exact-rescale branch on most tiles of many rows."""
import math
import os

import pytest
import torch

from _parity import assert_repeatable
from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
ROW_TOL = 1.2e-2      # per-row |o - ref| / |ref|: bf16 P (2^-9 per product, averaged) + the bf16 output rounding (2^-9) leave
                      # 3-4e-3 measured; 1.2e-2 is 3x that and an order of magnitude under one corrupted row (>= 0.2)


def _perm(n):
    i = torch.arange(n, device=DEV)
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def _make_vt(v, s_pad):
    B, S, D = v.shape
    vt = torch.zeros(B, D, s_pad, dtype=BF, device=DEV)
    vt[:, :, _perm(s_pad)[:S]] = v.transpose(1, 2)
    return vt


def _sdpa_rows_fp32(q, k, v, scale, causal=False, chunk=2048):
    """[Sq, 128] x [Skv, 128] -> fp32 softmax(q k^T scale) v, chunked over queries."""
    out = torch.empty(q.shape[0], v.shape[1], dtype=torch.float32, device=DEV)
    kf, vf = k.float(), v.float()
    for s in range(0, q.shape[0], chunk):
        sc = (q[s:s + chunk].float() @ kf.t()) * scale
        if causal:
            qi = torch.arange(s, min(s + chunk, q.shape[0]), device=DEV)[:, None]
            sc = sc.masked_fill(torch.arange(k.shape[0], device=DEV)[None, :] > qi, float("-inf"))
        out[s:s + chunk] = torch.softmax(sc, dim=-1) @ vf
    return out


def _check_rows(o, ref, what):
    err = (o.float() - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-3 * ref.norm(dim=-1).mean())
    worst = err.max().item()
    bad = int((err > ROW_TOL).sum().item())
    if os.environ.get("ALG_PARITY_REPORT"):
        import json
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_floor.jsonl"), "a") as f:
            f.write(json.dumps({"case": what, "rows": int(err.numel()), "row_err_max": worst,
                                "row_err_mean": err.mean().item(), "row_tol": ROW_TOL}) + "\n")
    assert bad == 0, "%s: %d of %d rows exceed %.1e (worst %.3e at row %d)" % (what, bad, err.numel(), ROW_TOL, worst,
                                                                                int(err.argmax().item()))


@pytest.mark.parametrize("q64", ["pipe", "0", "1"])   # the pipelined 32-query kernel, attention128.hip, the 64-query kernel (default)
@pytest.mark.parametrize("name,S", [("c3_wan480p", 32760), ("c5_wan720p", 75600), ("c4_hunyuan720p", 118800 + 256)])
def test_flash_attn_d128_every_row_vs_fp32_sdpa_at_full_s(name, S, q64, monkeypatch, request):
    monkeypatch.setenv("ALG_ATTN128_PIPE", "1" if q64 == "pipe" else "0")
    monkeypatch.setenv("ALG_ATTN128_Q64", "1" if q64 == "1" else "0")   # "1": the 64-query kernel, the default at these lengths
    H = 3
    D = H * 128
    g = torch.Generator(device=DEV).manual_seed(S)
    q = (torch.randn(1, S, D, generator=g, device=DEV) * 2.5).to(BF)        # scores ~ N(0, 2.5^2): peaked rows
    k = torch.randn(1, S, D, generator=g, device=DEV).to(BF)
    v = torch.randn(1, S, D, generator=g, device=DEV).to(BF)
    k[0, S // 3] *= 3.0                                                      # a dominant key far from the first tile
    q[0, 5::97] *= 0.05                                                      # and some near-uniform rows (O ~ mean of V)
    s_pad = (S + 63) // 64 * 64
    vt = _make_vt(v, s_pad)
    scale = 1.0 / math.sqrt(128)
    o = torch.full((1, S, D), 7.0, dtype=BF, device=DEV)
    call = lambda: _lib.flash_attn_d128(q, k, vt, o, 1, H, S, S, S * D, D, S * D, D, D * s_pad, s_pad, S * D, D, scale).clone()
    got = assert_repeatable(call, 8, "%s q64=%s" % (name, q64))
    for h in range(H):
        sl = slice(h * 128, (h + 1) * 128)
        ref = _sdpa_rows_fp32(q[0, :, sl], k[0, :, sl], v[0, :, sl], scale)
        _check_rows(got[0, :, sl], ref, "attn128_%s_q64=%s_head%d" % (name, q64, h))


@pytest.mark.parametrize("L", [1000, 577 + 256 + 103])
def test_flash_attn_d128_grouped_causal_every_row_at_llava_length(L):
    """The Llava-Llama-3 launch (`text_encoder_llava.py`): 32 query heads on 8 KV heads, causal, fused q|k rows."""
    H, Hk = 8, 2                                                             # same 4:1 grouping, a quarter of the heads
    D, KV = H * 128, Hk * 128
    g = torch.Generator(device=DEV).manual_seed(L)
    qk = (torch.randn(1, L, D + KV, generator=g, device=DEV) * 1.5).to(BF)
    v = torch.randn(1, L, KV, generator=g, device=DEV).to(BF)
    l_pad = (L + 63) // 64 * 64
    vt = _make_vt(v, l_pad)
    o = torch.full((1, L, D), 7.0, dtype=BF, device=DEV)
    call = lambda: _lib.flash_attn_d128(qk, qk, vt, o, 1, H, L, L, L * (D + KV), D + KV, L * (D + KV), D + KV, KV * l_pad,
                                        l_pad, L * D, D, 128 ** -0.5, k_off=D, kv_group=H // Hk, causal=True).clone()
    got = assert_repeatable(call, 8, "llava causal L=%d" % L)
    for h in range(H):
        kh = h // (H // Hk)
        ref = _sdpa_rows_fp32(qk[0, :, h * 128:(h + 1) * 128], qk[0, :, D + kh * 128:D + (kh + 1) * 128],
                              v[0, :, kh * 128:(kh + 1) * 128], 128 ** -0.5, causal=True)
        _check_rows(got[0, :, h * 128:(h + 1) * 128], ref, "attn128_llava_L%d_head%d" % (L, h))


def test_flash_attn_d64_every_row_vs_fp32_sdpa_at_c2_length():
    """The headline's launch shape (17,776 tokens, fused q|k rows, 2 CFG samples) on 6 of the 48 heads: every row."""
    S, H, nb = 17776, 6, 2
    Dh = H * 64
    g = torch.Generator(device=DEV).manual_seed(17)
    qk = (torch.randn(nb, S, 2 * Dh, generator=g, device=DEV) * 1.6).to(BF)
    v = torch.randn(nb, S, Dh, generator=g, device=DEV).to(BF)
    s_pad = (S + 63) // 64 * 64
    vt = _make_vt(v, s_pad)
    o = torch.full((nb, S, Dh), 7.0, dtype=BF, device=DEV)
    def call():
        o.fill_(7.0)
        _lib.flash_attn_d64(qk, qk, vt, o, nb, H, S, S * 2 * Dh, 2 * Dh, Dh * s_pad, s_pad, S * Dh, Dh, 0.125, k_off=Dh)
        return o.clone()

    got = assert_repeatable(call, 8, "d64 C2 launch")
    for b in range(nb):
        for h in (0, 3, 5):
            sl = slice(h * 64, (h + 1) * 64)
            ref = _sdpa_rows_fp32(qk[b, :, sl], qk[b, :, Dh + h * 64:Dh + (h + 1) * 64], v[b, :, sl], 0.125)
            _check_rows(got[b, :, sl], ref, "attn64_c2_sample%d_head%d" % (b, h))


@pytest.mark.parametrize("pp", ["7", "4", "0"])   # the 8-wave statement on 16x16x32 MFMAs (round 6), on 32x32x16 (default), the straight loop
def test_flash_attn_d64_prescaled_every_row_vs_fp32_sdpa_at_c2_length(pp, monkeypatch):
    """The PRODUCT's form of the headline launch -- Q carries scale * log2(e) (alg_qk_norm_rope_scaled), ALG_ATTN_Q_PRESCALED, the
    split-KV tail next to the main launch -- 17,776 tokens, 2 CFG samples, 8 heads (one per XCD: the tail plan engages), every row
    of 3 heads against fp32 SDPA of the same rounded Q, for each main-launch kernel; 8 runs bit-equal."""
    monkeypatch.setenv("ALG_ATTN_PP", pp)
    S, H, nb = 17776, 8, 2
    Dh = H * 64
    g = torch.Generator(device=DEV).manual_seed(19)
    qk = torch.randn(nb, S, 2 * Dh, generator=g, device=DEV).to(BF)
    qk[:, :, :Dh] = (qk[:, :, :Dh].float() * (0.125 * 1.4426950408889634)).to(BF)       # scores ~ N(0, 1.44^2) log2 units
    v = torch.randn(nb, S, Dh, generator=g, device=DEV).to(BF)
    s_pad = (S + 127) // 128 * 128
    vt = _make_vt(v, s_pad)
    o = torch.full((nb, S, Dh), 7.0, dtype=BF, device=DEV)

    def call():
        o.fill_(7.0)
        _lib.flash_attn_d64(qk, qk, vt, o, nb, H, S, S * 2 * Dh, 2 * Dh, Dh * s_pad, s_pad, S * Dh, Dh, 0.125, k_off=Dh, q_prescaled=True)
        return o.clone()

    got = assert_repeatable(call, 8, "d64 prescaled C2 launch, ALG_ATTN_PP=" + pp)
    for b in range(nb):
        for h in (0, 4, 7):
            sl = slice(h * 64, (h + 1) * 64)
            # scores are in log2 units already: softmax base 2 = softmax of s * ln 2
            ref = _sdpa_rows_fp32(qk[b, :, sl], qk[b, :, Dh + h * 64:Dh + (h + 1) * 64], v[b, :, sl], 0.6931471805599453)
            _check_rows(got[b, :, sl], ref, "attn64_c2_prescaled_pp%s_sample%d_head%d" % (pp, b, h))
