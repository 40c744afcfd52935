"""GPU parity of the DiT building-block kernels (C ABI) against plain fp32 torch restatements on CPU."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from alg_amd import _lib
from oracle import dit_oracle

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(BF)


def swap23(n):
    return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1)


def rel_err(got, ref):
    return ((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=["10", "9", "6"], ids=["4waves-asm-loop-16x16x32", "4waves-asm-loop", "pingpong-halftiles"])
def gemm_pipe(request, monkeypatch):
    """Every GEMM test runs on the three built schedules (10 = schedule 9's loop on v_mfma_f32_16x16x32_bf16, round 6; 9 = the
    asm loop on 32x32x16, round 3; 6 = the fp8-small-K / convolution schedule and 9's bit-identity reference)."""
    monkeypatch.setenv("ALG_GEMM_PIPE", request.param)
    return request.param


def _variant(request, monkeypatch, variant):
    """ALG_ATTN_VARIANT: 33 (default) and 1 (exact running max)."""
    assert variant in ("1", "33")
    monkeypatch.setenv("ALG_ATTN_VARIANT", variant)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 256), (300, 520, 128), (17, 64, 512), (2, 1000, 64),
                                   (1111, 96, 3072), (70, 250, 192), (33, 6, 64)])
def test_gemm_plain_bias(device, gemm_pipe, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A, B, bias = rnd((M, K), g), rnd((N, K), g, 0.1), rnd((N,), g)
    C = torch.full((M, N), 7.0, dtype=BF, device=device)
    _lib.gemm(A.to(device), B.to(device), C, M, N, K, K, K, N, bias=bias.to(device))
    ref = A.float() @ B.float().t() + bias.float()
    got = C.float().cpu()
    assert rel_err(got, ref) < 4e-3
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max()


def test_gemm_is_transpose_detecting(device, gemm_pipe):
    """A = I with an asymmetric B: catches a swapped C/D fragment layout (guide rule 16)."""
    M = N = K = 256
    A = torch.eye(M).to(BF)
    B = (torch.arange(N)[:, None] * 0.25 + torch.arange(K)[None, :] * 0.001953125).to(BF)
    C = torch.empty(M, N, dtype=BF, device=device)
    _lib.gemm(A.to(device), B.to(device), C, M, N, K, K, K, N)
    assert torch.equal(C.cpu().float(), B.float().t())


@pytest.mark.parametrize("act", [_lib.ACT_GELU_TANH, _lib.ACT_SILU])
def test_gemm_activation(device, gemm_pipe, act):
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 512, 192
    A, B, bias = rnd((M, K), g), rnd((N, K), g, 0.2), rnd((N,), g)
    C = torch.empty(M, N, dtype=BF, device=device)
    _lib.gemm(A.to(device), B.to(device), C, M, N, K, K, K, N, bias=bias.to(device), act=act)
    lin = (A.float() @ B.float().t() + bias.float()).to(BF).float()
    ref = F.gelu(lin, approximate="tanh") if act == _lib.ACT_GELU_TANH else F.silu(lin)
    got = C.float().cpu()
    assert rel_err(got, ref) < 6e-3
    assert (got - ref).abs().max() <= 3e-2 * ref.abs().max()


def test_gemm_batched_gated_residual_in_place(device, gemm_pipe):
    """The out-projection / FF2 form: x <- x + gate[seg] * (A @ W^T + b), per-sample batch, text/video segments."""
    g = torch.Generator().manual_seed(9)
    nb, S, N, K, T = 2, 290, 512, 128, 26
    A, Wt, bias = rnd((nb, S, K), g), rnd((N, K), g, 0.2), rnd((N,), g)
    X = rnd((nb, S, N), g)
    gate = rnd((nb, 7, N), g)  # strideGate = 7*N, pairs live at [:, 2:4]
    Xd = X.to(device).clone()
    _lib.gemm(A.to(device), Wt.to(device), Xd, S, N, K, K, K, N, bias=bias.to(device), R=Xd, ldr=N,
              gate=gate.to(device), gate_off=2 * N, strideGate=7 * N, seg_split=T, batch=nb, strideA=S * K,
              strideC=S * N, strideR=S * N)
    lin = (A.float() @ Wt.float().t() + bias.float()).to(BF)
    gsel = torch.where(torch.arange(S)[None, :, None] < T, gate[:, 2:3].float(), gate[:, 3:4].float())
    ref = (X.float() + (gsel * lin.float()).to(BF).float()).to(BF).float()
    got = Xd.float().cpu()
    assert rel_err(got, ref) < 5e-3
    assert (got - ref).abs().max() <= 3e-2 * ref.abs().max()


def test_gemm_transposed_v_projection(device, gemm_pipe):
    """V^T = Wv @ y^T with per-row bias and the bits-2/3 column permutation, pad columns untouched."""
    g = torch.Generator().manual_seed(11)
    nb, S, D = 2, 200, 256
    S_pad = 256
    y, Wv, bv = rnd((nb, S, D), g), rnd((D, D), g, 0.1), rnd((D,), g)
    vt = torch.zeros(nb, D, S_pad, dtype=BF, device=device)
    _lib.gemm(Wv.to(device), y.to(device), vt, D, S, D, D, D, S_pad, bias=bv.to(device), batch=nb, strideB=S * D,
              strideC=D * S_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
    v = y.float() @ Wv.float().t() + bv.float()  # [nb, S, D]
    ref = torch.zeros(nb, D, S_pad)
    perm = torch.tensor([swap23(n) for n in range(S)])
    ref[:, :, perm] = v.transpose(1, 2)
    got = vt.float().cpu()
    live = torch.zeros(S_pad, dtype=torch.bool)
    live[perm] = True
    assert torch.count_nonzero(got[:, :, ~live]) == 0
    assert rel_err(got, ref) < 4e-3


# ------------------------------------------------------------------------------------- attention
def run_attention(device, q, k, v, scale):
    """q,k,v [B, S, H, 64] bf16 (CPU) -> o [B, S, H, 64] via the kernel's strided layouts."""
    Bn, S, H, _ = q.shape
    D = H * 64
    S_pad = (S + 127) // 128 * 128  # the 128-kv-per-stage variants read one 64-row tile further
    qk = torch.cat([q.reshape(Bn, S, D), k.reshape(Bn, S, D)], dim=-1).contiguous().to(device)
    vt = torch.zeros(Bn, D, S_pad, dtype=BF)
    perm = torch.tensor([swap23(n) for n in range(S)])
    vt[:, :, perm] = v.reshape(Bn, S, D).transpose(1, 2)
    o = torch.zeros(Bn, S, D, dtype=BF, device=device)
    _lib.flash_attn_d64(qk, qk, vt.to(device), o, Bn, H, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, scale, k_off=D)
    return o.cpu().reshape(Bn, S, H, 64)


def sdpa_ref(q, k, v, scale):
    qq, kk, vv = (t.double().transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qq @ kk.transpose(-1, -2) * scale, dim=-1)
    return (p @ vv).transpose(1, 2)


@pytest.mark.parametrize("variant", ["1", "33"])   # 33 = the default (lazy running max), 1 = the exact-running-max reference
@pytest.mark.parametrize("Bn,S,H", [(1, 64, 1), (1, 100, 3), (2, 273, 9), (1, 1000, 8), (3, 994, 2), (1, 17, 1)])
def test_flash_attention_vs_sdpa(device, monkeypatch, request, Bn, S, H, variant):
    _variant(request, monkeypatch, variant)  # every kernel variant must pass, not just the default
    g = torch.Generator().manual_seed(S + H)
    q, k, v = rnd((Bn, S, H, 64), g), rnd((Bn, S, H, 64), g), rnd((Bn, S, H, 64), g)
    got = run_attention(device, q, k, v, 0.125).double()
    ref = sdpa_ref(q, k, v, 0.125)
    assert (got - ref).abs().max() <= 2e-2, (got - ref).abs().max()
    assert rel_err(got, ref) < 1e-2


@pytest.mark.parametrize("variant", ["1", "33"])   # 33 = the default (lazy running max), 1 = the exact-running-max reference
def test_flash_attention_forced_rescale_and_asymmetry(device, monkeypatch, request, variant):
    """A key that dominates late in the sequence forces the online-softmax rescale; V = one-hot rows make any
    kv-order / transpose mistake in the P@V operand layout visible."""
    _variant(request, monkeypatch, variant)
    g = torch.Generator().manual_seed(2)
    S = 320
    q, k = rnd((1, S, 1, 64), g), rnd((1, S, 1, 64), g)
    k[0, 257, 0] = q[0, 5, 0] * 4  # spike for query 5 at tile 4
    v = torch.zeros(1, S, 1, 64, dtype=BF)
    v[0, torch.arange(S), 0, torch.arange(S) % 64] = 1.0
    v[0, :, 0, 63] += (torch.arange(S) / S).to(BF)
    got = run_attention(device, q, k, v, 0.125).double()
    ref = sdpa_ref(q, k, v, 0.125)
    assert (got - ref).abs().max() <= 1e-2
    assert ref[0, 5, 0, 257 % 64] > 0.9  # the spike really dominates


@pytest.mark.parametrize("variant", ["1", "33"])  # 34 pre-scales q (one more bf16 rounding): fine at real score
# magnitudes (the other tests), not at the |score| ~ 220 this test drives
@pytest.mark.parametrize("gain", [0.5, 2.5, 3.4, 4.0, 6.5, 7.5, 40.0])   # scores ~ 11.5 x gain log2 units: 75 / 86 straddle 2^80
def test_flash_attention_lazy_max_thresholds(device, monkeypatch, request, variant, gain):
    """The default softmax keeps a LAZY running max: probabilities are formed against the current m and the exact
    max / rescale path only runs when a row sum leaves [0, 2^80).  Spikes that stay below the threshold (scores up to ~220
    above m: probabilities up to 2^39), cross it, or overflow exp2 outright (gain 40: +inf) must all give the softmax the
    reference gives -- also on a ragged last tile and with the spike in the first tile (m = -inf start)."""
    _variant(request, monkeypatch, variant)
    g = torch.Generator().manual_seed(7)
    S = 333
    q, k = rnd((1, S, 2, 64), g), rnd((1, S, 2, 64), g)
    k[0, 300, 0] = q[0, 5, 0] * gain       # late spike for query 5, head 0
    k[0, 3, 1] = q[0, 9, 1] * gain         # spike inside the first tile for query 9, head 1
    k[0, 330, 1] = q[0, 200, 1] * gain     # spike in the ragged tail tile
    v = rnd((1, S, 2, 64), g)
    got = run_attention(device, q, k, v, 0.125).double()
    ref = sdpa_ref(q, k, v, 0.125)
    assert torch.isfinite(got).all() and (got - ref).abs().max() <= 1.5e-2


# ------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("D,rows,T", [(512, 37, 5), (3072, 20, 7), (1024, 9, 0)])
def test_layernorm_modulate(device, D, rows, T):
    g = torch.Generator().manual_seed(D)
    nb = 2
    x, w, b = rnd((nb, rows, D), g, 2.0), (1 + 0.1 * torch.randn(D, generator=g)).to(BF), rnd((D,), g, 0.1)
    mod = rnd((nb, 6 * D), g, 0.5)
    y = torch.empty(nb, rows, D, dtype=BF, device=device)
    _lib.layernorm_modulate(x.to(device), y, w.to(device), b.to(device), mod.to(device), mod.to(device), 6 * D, nb,
                            rows, D, T, 1e-5, scale_off=2 * D, shift_off=0)
    n = F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5).to(BF)
    seg = (torch.arange(rows) >= T).long()
    shift = torch.stack([mod[:, :D], mod[:, D:2 * D]], 1)[:, seg]       # [nb, rows, D]
    scale = torch.stack([mod[:, 2 * D:3 * D], mod[:, 3 * D:4 * D]], 1)[:, seg]
    ref = (n * (1 + scale) + shift).float()
    got = y.float().cpu()
    assert (got - ref).abs().max() <= 4e-2 and rel_err(got, ref) < 6e-3
    # plain LayerNorm + strided batches (the final-norm form)
    y2 = torch.zeros(nb, rows - 2, D, dtype=BF, device=device)
    _lib.layernorm_modulate(x.to(device), y2, w.to(device), b.to(device), None, None, 0, nb, rows - 2, D, 0, 1e-5,
                            x_bstride=rows * D, y_bstride=(rows - 2) * D, x_off=2 * D)
    assert (y2.float().cpu() - n[:, 2:].float()).abs().max() <= 2e-2


@pytest.mark.parametrize("nb,rows,D,T", [(2, 17776, 3072, 226), (3, 3000, 1536, 226), (1, 4100, 512, 0), (2, 2100, 3072, 2099)])
def test_layernorm_modulate_multi_row_form_is_bit_identical(device, nb, rows, D, T):
    """From 4,096 rows up alg_layernorm_modulate runs ln_mod_rows_kernel (eight rows per wave, the four parameter rows held in
    registers, the next row's loads in flight): same operations per element, so the same bits as the one-row-per-wave kernel --
    which here processes the same tensors in slices of at most 2,048 rows (below the switch).  Row counts that leave a ragged
    last wave, the text -> video boundary inside a wave, a boundary at 0 and at rows - 1, three batch items."""
    g = torch.Generator(device=device).manual_seed(rows + D)
    x = (torch.randn(nb, rows, D, generator=g, device=device) * 1.5 + 0.3).to(BF)
    w, b = (1 + 0.1 * torch.randn(D, generator=g, device=device)).to(BF), (0.1 * torch.randn(D, generator=g, device=device)).to(BF)
    mod = (0.3 * torch.randn(nb, 6 * D, generator=g, device=device)).to(BF)
    y = torch.full((nb, rows, D), 5.0, dtype=BF, device=device)
    _lib.layernorm_modulate(x, y, w, b, mod, mod, 6 * D, nb, rows, D, T, 1e-5, scale_off=2 * D, shift_off=0)
    want = torch.full((nb, rows, D), 7.0, dtype=BF, device=device)
    for bi in range(nb):
        for a in range(0, rows, 2048):
            n = min(2048, rows - a)
            _lib.layernorm_modulate(x, want, w, b, mod, mod, 6 * D, 1, n, D, max(0, min(T - a, n)), 1e-5,
                                    x_off=(bi * rows + a) * D, y_off=(bi * rows + a) * D, scale_off=bi * 6 * D + 2 * D,
                                    shift_off=bi * 6 * D)
    assert torch.equal(y, want)


def test_qk_norm_rope(device):
    g = torch.Generator().manual_seed(4)
    nb, S, H, T = 2, 50, 3, 10
    cfg = dit_oracle.DiTConfig(sample_height=8, sample_width=10, sample_frames=5)
    cos, sin = dit_oracle.rope_tables(cfg, 64, 80, 2)  # 2 * 4 * 5 = 40 video tokens
    assert cos.shape == (S - T, 64)
    qk = rnd((nb, S, 2, H, 64), g, 1.5)
    wq, bq, wk, bk = [(1 + 0.1 * torch.randn(64, generator=g)).to(BF) if i % 2 == 0 else rnd((64,), g, 0.1)
                      for i in range(4)]
    d = qk.to(device).clone()
    _lib.qk_norm_rope_(d, wq.to(device), bq.to(device), wk.to(device), bk.to(device), cos.to(device), sin.to(device),
                       nb, S, H, T, 1e-6)
    q = F.layer_norm(qk[:, :, 0].float(), (64,), wq.float(), bq.float(), 1e-6).to(BF).transpose(1, 2)  # [nb,H,S,64]
    k = F.layer_norm(qk[:, :, 1].float(), (64,), wk.float(), bk.float(), 1e-6).to(BF).transpose(1, 2)
    q = torch.cat([q[:, :, :T], dit_oracle.apply_rotary(q[:, :, T:], cos, sin)], dim=2)
    k = torch.cat([k[:, :, :T], dit_oracle.apply_rotary(k[:, :, T:], cos, sin)], dim=2)
    ref = torch.stack([q.transpose(1, 2), k.transpose(1, 2)], dim=2).float()
    got = d.float().cpu()
    assert (got - ref).abs().max() <= 4e-2 and rel_err(got, ref) < 5e-3


def test_qk_norm_rope_scaled_and_prescaled_attention(device):
    """alg_qk_norm_rope_scaled folds softmax_scale * log2(e) into Q's last rounding (K untouched); the attention entry with
    ALG_ATTN_Q_PRESCALED then takes the scores as log2 units.  Q(scaled) == c * Q(unscaled) to one bf16 rounding, K bit
    equal, and the prescaled attention gives the reference softmax on the ORIGINAL q, k (also across the split-KV tail
    threshold shapes and a ragged sequence)."""
    g = torch.Generator().manual_seed(8)
    nb, S, H, T = 2, 50, 3, 10
    cfg = dit_oracle.DiTConfig(sample_height=8, sample_width=10, sample_frames=5)
    cos, sin = dit_oracle.rope_tables(cfg, 64, 80, 2)
    qk = rnd((nb, S, 2, H, 64), g, 1.5)
    ws = [(1 + 0.1 * torch.randn(64, generator=g)).to(BF) if i % 2 == 0 else rnd((64,), g, 0.1) for i in range(4)]
    args = [w.to(device) for w in ws] + [cos.to(device), sin.to(device), nb, S, H, T, 1e-6]
    a, b = qk.to(device).clone(), qk.to(device).clone()
    c = 0.125 * 1.4426950408889634
    _lib.qk_norm_rope_(a, *args)
    _lib.qk_norm_rope_(b, *args, q_scale=c)
    assert torch.equal(a[:, :, 1], b[:, :, 1])                                     # K untouched
    qa, qb = a[:, :, 0].float(), b[:, :, 0].float()
    assert ((qb - qa * c).abs() <= 2.0 ** -7 * (qa * c).abs() + 1e-6).all()      # one bf16 rounding each: <= 2 * 2^-9 apart
    for Bn, S2, H2 in [(1, 333, 2), (2, 1000, 3)]:
        q, k, v = rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g)
        ref = sdpa_ref(q, k, v, 0.125)
        D = H2 * 64
        S_pad = (S2 + 127) // 128 * 128
        qs = (q.float() * c).to(BF)                                              # what the scaled norm kernel would hand over
        qkb = torch.cat([qs.reshape(Bn, S2, D), k.reshape(Bn, S2, D)], dim=-1).contiguous().to(device)
        vt = torch.zeros(Bn, D, S_pad, dtype=BF)
        vt[:, :, torch.tensor([swap23(n) for n in range(S2)])] = v.reshape(Bn, S2, D).transpose(1, 2)
        o = torch.zeros(Bn, S2, D, dtype=BF, device=device)
        _lib.flash_attn_d64(qkb, qkb, vt.to(device), o, Bn, H2, S2, S2 * 2 * D, 2 * D, D * S_pad, S_pad, S2 * D, D, 0.125,
                            k_off=D, q_prescaled=True)
        got = o.cpu().reshape(Bn, S2, H2, 64).double()
        assert torch.isfinite(got).all() and (got - ref).abs().max() <= 1e-2


@pytest.mark.parametrize("Bn,S2,H2", [(1, 512, 2), (2, 1000, 3), (1, 513, 1), (1, 575, 2), (1, 640, 1), (1, 700, 2), (1, 832, 1),
                                       (1, 2050, 2), (8, 1200, 1)])
def test_flash_attention_d64_m16_kernel(device, monkeypatch, Bn, S2, H2):
    """attention64_m16.hip (round 6: the 8-wave statement kernel on v_mfma_f32_16x16x32_bf16 -- 16 x 16 score blocks, two queries per
    lane, the K-fragment row map that makes a lane's S blocks the PV operand in V^T's stored order; ALG_ATTN_PP=7 for pre-scaled calls
    over >= 12 KV tiles) against fp32 SDPA and against attention.hip's 32x32x16 statement kernel (ALG_ATTN_PP=4) on the same
    tensors: ragged tails (the 16 x 16 layout's own key mask), tile counts that leave 0-3 tiles behind the statement's groups of
    four, rows whose first-tile max is beyond +-64 (non-zero offset: those waves never enter the statement) next to rows in the
    zero-offset form, late dominant keys (the statement refuses the tile, the exact path runs), query blocks ending mid-wave; 8 x 1
    heads = enough units for the split-KV tail plan to engage next to the m16 main launch.  Run-to-run identical."""
    g = torch.Generator().manual_seed(S2 + H2)
    c = 0.125 * 1.4426950408889634
    q, k, v = rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g)
    q[:, : S2 // 3] *= 9.0                       # scores ~ +-70 in log2 units: non-zero offsets for a third of the rows
    k[:, (2 * S2) // 3] *= 6.0                   # one late key that dominates
    D = H2 * 64
    S_pad = (S2 + 127) // 128 * 128
    qs = (q.float() * c).to(BF)
    # the reference takes the SAME rounded, pre-scaled Q the kernels see (scores in log2 units): with scores in the hundreds
    # the rounding of q * c alone moves probabilities by tens of percent, which is not what this test is about
    sc2 = torch.einsum("bqhd,bkhd->bhqk", qs.double(), k.double()) * math.log(2.0)
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc2, dim=-1), v.double())
    qkb = torch.cat([qs.reshape(Bn, S2, D), k.reshape(Bn, S2, D)], dim=-1).contiguous().to(device)
    vt = torch.zeros(Bn, D, S_pad, dtype=BF)
    vt[:, :, torch.tensor([swap23(n) for n in range(S2)])] = v.reshape(Bn, S2, D).transpose(1, 2)
    vt = vt.to(device)
    outs, errs = {}, {}
    for flag in ("7", "4"):
        monkeypatch.setenv("ALG_ATTN_PP", flag)
        o = torch.full((Bn, S2, D), 3.0, dtype=BF, device=device)
        _lib.flash_attn_d64(qkb, qkb, vt, o, Bn, H2, S2, S2 * 2 * D, 2 * D, D * S_pad, S_pad, S2 * D, D, 0.125, k_off=D,
                            q_prescaled=True)
        outs[flag] = o
        got = o.cpu().reshape(Bn, S2, H2, 64).double()
        assert torch.isfinite(got).all(), flag
        errs[flag] = ((got - ref).abs().max().item(), (got - ref).abs().mean().item())
    assert errs["4"][0] <= 3e-2 and errs["4"][1] <= 2e-3, errs
    assert errs["7"][0] <= 3e-2 and errs["7"][1] <= 2e-3, errs
    assert (outs["7"].float() - outs["4"].float()).abs().max().item() <= 3.2e-2
    monkeypatch.setenv("ALG_ATTN_PP", "7")
    o2 = torch.empty_like(outs["7"])
    _lib.flash_attn_d64(qkb, qkb, vt, o2, Bn, H2, S2, S2 * 2 * D, 2 * D, D * S_pad, S_pad, S2 * D, D, 0.125, k_off=D,
                        q_prescaled=True)
    assert torch.equal(o2, outs["7"])


def test_patchify_unpatchify_timestep(device):
    g = torch.Generator().manual_seed(6)
    n, Fr, C, H, W, p = 3, 2, 4, 6, 10, 2
    lat = rnd((1, Fr, C, H, W), g)
    conds = [rnd((1, Fr, C, H, W), g) for _ in range(n)]
    out = torch.empty(n, Fr * (H // p) * (W // p), 2 * C * p * p, dtype=BF, device=device)
    _lib.patchify(lat.to(device), 0, [c.to(device) for c in conds], out, n, Fr, C, H, W, p)
    for i in range(n):
        x = torch.cat([lat[0], conds[i][0]], dim=1).float()  # [F, 2C, H, W]  (cog:1068-1070 channel concat)
        ref = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 2 * C * p * p)  # conv2d im2col order
        assert torch.equal(out[i].float().cpu(), ref)
    tok = rnd((n, Fr * (H // p) * (W // p), C * p * p), g)
    back = torch.empty(n, Fr, C, H, W, dtype=BF, device=device)
    _lib.unpatchify(tok.to(device), back, n, Fr, C, H, W, p)
    ref = tok.reshape(n, Fr, H // p, W // p, C, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    assert torch.equal(back.cpu(), ref)
    t = torch.tensor([999.0, 19.0, 500.0])
    emb = torch.empty(3, 512, dtype=BF, device=device)
    _lib.timestep_embedding(t.to(device), emb, 3, 512, True)
    ref = dit_oracle.timestep_sinusoid(t, 512, True, 0)
    assert (emb.float().cpu() - ref).abs().max() <= 8e-3  # bf16 output of values in [-1, 1]


def test_pingpong_gemm_race_screen(monkeypatch):
    """The default GEMM schedule keeps LDS-DMA loads in flight across barriers (counted vmcnt): an ordering bug would show
    as rare wrong tiles.  Its output must equal the drain-and-barrier schedule bit for bit (same K order), over random
    shapes and repeated runs of a large one."""
    import random
    rng = random.Random(1)
    g = torch.Generator(device="cuda").manual_seed(1)
    cases = [(rng.randint(1, 3000), rng.randint(1, 300) * 4, rng.randint(1, 48) * 64) for _ in range(40)]
    cases += [(17776, 3072, 3072)] * 4
    for M, N, K in cases:
        a = torch.randn(M, K, generator=g, device="cuda").to(BF)
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(BF)
        outs = {}
        base = "6"          # the 8-wave ping-pong schedule: the bit-identity reference
        for pipe in (base, "6", "6", "9", "9", "9"):
            monkeypatch.setenv("ALG_GEMM_PIPE", pipe)
            c = torch.empty(M, N, dtype=BF, device="cuda")
            _lib.gemm(a, w, c, M, N, K, K, K, N)
            if pipe in outs:
                assert torch.equal(c, outs[base]), (M, N, K)
            outs.setdefault(pipe, c)
        assert torch.equal(outs["6"], outs[base]) and torch.equal(outs["9"], outs[base]), (M, N, K)
        # schedule 10 (the default since round 6) sums 32 products per MFMA: other rounding points, so its screen is run-to-run
        # equality plus one bf16 step from the reference
        monkeypatch.setenv("ALG_GEMM_PIPE", "10")
        c10 = [torch.empty(M, N, dtype=BF, device="cuda") for _ in range(3)]
        for c in c10:
            _lib.gemm(a, w, c, M, N, K, K, K, N)
        assert torch.equal(c10[0], c10[1]) and torch.equal(c10[0], c10[2]), (M, N, K)
        d = (c10[0].float() - outs[base].float()).abs()
        assert bool((d <= 2.0 ** -7 * outs[base].float().abs() + 1e-3 * outs[base].float().abs().max()).all()), (M, N, K)


@pytest.mark.parametrize("S,D,N", [(17776, 3072, 2), (300, 512, 3), (1000, 256, 1), (257, 128, 2)])
def test_gemm_pair_is_bit_identical_to_the_two_launches(monkeypatch, S, D, N):
    """alg_gemm_bf16_pair: a block's Q|K projection and its transposed, permuted V projection (different operand roles, the
    same activations) as ONE persistent launch -- every tile computed as by its own launch, so the bits are those of the two
    separate calls, at the C2 shape and at shapes with edge tiles in both problems; under ALG_GEMM_PIPE=6 and for calls the pair
    form cannot take (K = 64; a residual) it is the two launches one after the other; bad arguments are rejected before any
    launch."""
    monkeypatch.setenv("ALG_GEMM_PIPE", "9")   # (schedule 10's pair tests: test_gpu_gemm_p10.py; 9 and 6 share their bits)
    g = torch.Generator(device="cuda").manual_seed(S + D)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    S_pad = (S + 63) // 64 * 64
    y, wqk, bqk, wv, bv = rn(N, S, D), rn(2 * D, D, sc=0.05), rn(2 * D), rn(D, D, sc=0.05), rn(D)

    def calls(qk, vt, K=D):
        return (((y, wqk, qk, S, 2 * D, K, D, D, 2 * D), dict(bias=bqk, batch=N, strideA=S * D, strideC=S * 2 * D)),
                ((wv, y, vt, D, S, K, D, D, S_pad), dict(bias=bv, batch=N, strideB=S * D, strideC=D * S_pad,
                                                        flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)))

    def separate(K=D):
        qk, vt = torch.full((N, S, 2 * D), 7.0, dtype=BF, device="cuda"), torch.zeros(N, D, S_pad, dtype=BF, device="cuda")
        for a, kw in calls(qk, vt, K):
            _lib.gemm(*a, **kw)
        return qk, vt

    def paired(K=D):
        qk, vt = torch.full((N, S, 2 * D), 7.0, dtype=BF, device="cuda"), torch.zeros(N, D, S_pad, dtype=BF, device="cuda")
        _lib.gemm_pair(*calls(qk, vt, K))
        return qk, vt

    want = separate()
    for _ in range(3):
        got = paired()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.count_nonzero(got[1][:, :, (S + 15) // 16 * 16:]) == 0   # pad columns of V^T (beyond S's 16-column permutation block) stay zero
    if D >= 128:                                                           # K = 64: schedule 9 cannot take it -> two launches
        w64, g64 = separate(64), paired(64)
        assert torch.equal(g64[0], w64[0]) and torch.equal(g64[1], w64[1])
    monkeypatch.setenv("ALG_GEMM_PIPE", "6")
    got = paired()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    monkeypatch.setenv("ALG_GEMM_PIPE", "9")
    qk = torch.zeros(N, S, 2 * D, dtype=BF, device="cuda")
    bad = (((y, wqk, qk, S, 2 * D, D - 8, D, D, 2 * D), dict(batch=N, strideA=S * D, strideC=S * 2 * D)), calls(qk, qk)[0])
    with pytest.raises(_lib.AlgHipError):                                   # K % 64 != 0 in the FIRST problem: nothing launched
        _lib.gemm_pair(bad[0], bad[1])
    with pytest.raises(_lib.AlgHipError):                                   # ... or in the second
        _lib.gemm_pair(bad[1], bad[0])


@pytest.mark.parametrize("S,heads,N,T,rope,qs", [(17776, 48, 2, 226, True, 0.18033688), (300, 8, 3, 17, True, 1.0),
                                                  (1000, 4, 1, 0, True, 0.18033688), (257, 12, 2, 257, True, 1.0),
                                                  (530, 4, 2, 100, False, 0.18033688), (300, 6, 1, 17, True, 0.18033688),
                                                  (300, 16, 2, 17, True, 0.18033688), (130, 32, 1, 0, False, 1.0)])   # (2 * heads % 32 == 0: one wave per token)
def test_gemm_pair_qk_is_bit_identical_to_the_pair_launch_plus_qk_norm_rope(monkeypatch, S, heads, N, T, rope, qs):
    """alg_gemm_bf16_pair_qk: the per-head QK LayerNorm + rotary embedding (+ softmax scale on Q) inside the Q|K projection's
    store loop.  Same arithmetic, same reduction tree, same rounding points as qk_norm_rope_kernel (csrc/qk_norm_rope.h), so
    the bits are those of alg_gemm_bf16_pair followed by alg_qk_norm_rope_scaled: at the C2 shape, with edge tiles in M, text /
    video boundaries inside a 16-row group, no text tokens, only text tokens, no rotary tables, and (heads = 6: a tile would
    straddle Q | K) through the documented fallback; V^T is untouched by the fusion; ALG_GEMM_PIPE=6 takes the fallback too."""
    monkeypatch.setenv("ALG_GEMM_PIPE", "9")   # (schedule 10: test_gpu_gemm_p10.py)
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(S + heads)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    S_pad = (S + 63) // 64 * 64
    y, wqk, bqk, wv, bv = rn(N, S, D), rn(2 * D, D, sc=0.05), rn(2 * D), rn(D, D, sc=0.05), rn(D)
    wq, bq, wk, bk = (1 + rn(64, sc=0.2)), rn(64, sc=0.2), (1 + rn(64, sc=0.2)), rn(64, sc=0.2)
    ang = torch.rand(max(S - T, 1), 32, generator=g, device="cuda") * 6.28
    cos = ang.cos().repeat_interleave(2, dim=1).contiguous() if rope else None
    sin = ang.sin().repeat_interleave(2, dim=1).contiguous() if rope else None

    def calls(qk, vt):
        return (((y, wqk, qk, S, 2 * D, D, D, D, 2 * D), dict(bias=bqk, batch=N, strideA=S * D, strideC=S * 2 * D)),
                ((wv, y, vt, D, S, D, D, D, S_pad), dict(bias=bv, batch=N, strideB=S * D, strideC=D * S_pad,
                                                        flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)))

    def fresh():
        return torch.full((N, S, 2 * D), 7.0, dtype=BF, device="cuda"), torch.zeros(N, D, S_pad, dtype=BF, device="cuda")

    qk0, vt0 = fresh()
    _lib.gemm_pair(*calls(qk0, vt0))
    raw = qk0.clone()
    _lib.qk_norm_rope_(qk0, wq, bq, wk, bk, cos, sin, N, S, heads, T, 1e-6, q_scale=qs)
    assert not torch.equal(raw, qk0)
    for _ in range(3):
        qk1, vt1 = fresh()
        _lib.gemm_pair_qk(*calls(qk1, vt1), wq, bq, wk, bk, cos, sin, heads, T, 1e-6, q_scale=qs)
        assert torch.equal(vt1, vt0)
        if not torch.equal(qk1, qk0):
            d = (qk1.float() - qk0.float()).abs()
            idx = (d > 0).nonzero()
            raise AssertionError("fused != separate: %d elements, max %.4g, first %s" % (idx.shape[0], d.max().item(), idx[:4].tolist()))
    monkeypatch.setenv("ALG_GEMM_PIPE", "6")
    qk1, vt1 = fresh()
    _lib.gemm_pair_qk(*calls(qk1, vt1), wq, bq, wk, bk, cos, sin, heads, T, 1e-6, q_scale=qs)
    assert torch.equal(qk1, qk0) and torch.equal(vt1, vt0)
    monkeypatch.setenv("ALG_GEMM_PIPE", "9")
    with pytest.raises(_lib.AlgHipError):                                   # not the [S][2][heads][64] layout: rejected, nothing launched
        _lib.gemm_pair_qk(*calls(qk1, vt1), wq, bq, wk, bk, cos, sin, heads + 1, T, 1e-6)
    with pytest.raises(_lib.AlgHipError):                                   # cos without sin
        _lib.gemm_pair_qk(*calls(qk1, vt1), wq, bq, wk, bk, ang, None, heads, T, 1e-6)


@pytest.mark.parametrize("form", ["plain", "gelu", "vt", "res", "res_gate_seg", "res_gate_f32", "res_gate_f32_straddle"])
def test_schedule9_equals_the_drain_and_barrier_schedule_bit_for_bit(monkeypatch, form):
    """Schedule 9 (hand-written asm K loop, accumulators in AGPRs, residual quads fetched INSIDE the loop over its first eight
    steady-state k-tiles, the rest by a catch-up chain) against schedule 0 (compiler-scheduled, drain + barrier per k-tile):
    same accumulation order, so equal bits.  K / 64 = 2 .. 13 enters the catch-up chain at each of its labels (0 .. 8 k-tiles
    short of the eight fetching ones) and runs past it; the shapes have edge tiles in M and N; every store-loop variant
    (no gate, bf16 gate with a per-row segment select, fp32 gate on one segment, the generic loop for a straddled tile)."""
    g = torch.Generator(device="cuda").manual_seed(9)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    for M, N, K in [(300, 520, 64 * k) for k in range(2, 14)] + [(1111, 96, 3072), (2100, 1024, 64 * 23)]:
        a, w, bias, x0 = rn(M, K), rn(N, K, sc=0.05), rn(N), rn(M, N)
        gate, gate32, brow = rn(1, 2 * N, sc=0.5), torch.randn(1, 2 * N, generator=g, device="cuda"), rn(M)

        def run():
            if form == "vt":
                npad = (N + 63) // 64 * 64
                c = torch.zeros(M, npad, dtype=BF, device="cuda")
                _lib.gemm(a, w, c, M, N, K, K, K, npad, bias=brow, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
                return c
            if form in ("plain", "gelu"):
                c = torch.full((M, N), 7.0, dtype=BF, device="cuda")
                _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH if form == "gelu" else _lib.ACT_NONE)
                return c
            x = x0.clone()
            kw = {}
            if form == "res_gate_seg":
                kw = dict(gate=gate, strideGate=2 * N, seg_split=M // 3)
            elif form == "res_gate_f32":
                kw = dict(gate=gate32, strideGate=2 * N, seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)
            elif form == "res_gate_f32_straddle":
                kw = dict(gate=gate32, strideGate=2 * N, seg_split=100, flags=_lib.GEMM_GATE_F32)
            _lib.gemm(a, w, x, M, N, K, K, K, N, bias=bias, R=x, ldr=N, **kw)
            return x

        monkeypatch.setenv("ALG_GEMM_PIPE", "6")   # the ping-pong schedule
        want = run()
        monkeypatch.setenv("ALG_GEMM_PIPE", "9")
        for _ in range(2):
            assert torch.equal(run(), want), (form, M, N, K)


@pytest.mark.parametrize("std", [4.0, 8.0, 15.0, 22.0])
def test_attention_wide_score_ranges_against_fp64(device, std, monkeypatch):
    """Round 4 raised the lazy running max's row-sum limit from 2^40 to 2^80 (common.h): rows stay on the no-rescale path with
    probabilities up to 2^80.  Scores ~ N(0, std^2) in log2 units over 2,050 keys put the row maxima at ~ 3.5 std: well inside
    (4, 8: the regime that used to throw a quarter of the waves out of the pipelined statement), around (15, 22) and beyond the
    limit (rows cross it mid-sequence: exact path, offsets stop being zero).  d = 64 (pre-scaled, pipelined main launch) and
    d = 128 (the pipelined 32-query kernel, and the 64-query kernel forced onto this length: its limit is 2^40 with an in-line
    exact path), every row against an fp64 softmax of the same bf16 inputs."""
    g = torch.Generator().manual_seed(int(std))
    S, H = 2050, 2
    for hd, call in ((64, "d64"), (128, "d128"), (128, "d128_q64")):
        monkeypatch.setenv("ALG_ATTN128_Q64", "2" if call == "d128_q64" else "0")
        q = (torch.randn(1, S, H, hd, generator=g) * (std / hd ** 0.5)).to(BF)      # q . k ~ N(0, std^2)
        k, v = rnd((1, S, H, hd), g), rnd((1, S, H, hd), g)
        D = H * hd
        S_pad = (S + 127) // 128 * 128
        vt = torch.zeros(1, D, S_pad, dtype=BF)
        vt[:, :, torch.tensor([swap23(n) for n in range(S)])] = v.reshape(1, S, D).transpose(1, 2)
        o = torch.full((1, S, D), 3.0, dtype=BF, device=device)
        if call == "d64":
            qkb = torch.cat([q.reshape(1, S, D), k.reshape(1, S, D)], dim=-1).contiguous().to(device)
            _lib.flash_attn_d64(qkb, qkb, vt.to(device), o, 1, H, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D,
                                q_prescaled=True)
            logits = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * math.log(2.0)
        else:
            scale = math.log(2.0)                                                   # (q . k) * scale * log2(e) = q . k log2 units
            _lib.flash_attn_d128(q.reshape(1, S, D).to(device), k.reshape(1, S, D).to(device), vt.to(device), o, 1, H, S, S,
                                 S * D, D, S * D, D, D * S_pad, S_pad, S * D, D, scale)
            logits = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * math.log(2.0)
        ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(logits, dim=-1), v.double())
        got = o.cpu().reshape(1, S, H, hd).double()
        assert torch.isfinite(got).all(), (call, std)
        err = (got - ref).abs()
        assert err.max().item() <= 3e-2 and err.mean().item() <= 2e-3, (call, std, err.max().item(), err.mean().item())


@pytest.mark.parametrize("Bn,S2,H2", [(1, 512, 2), (2, 1000, 3), (1, 513, 1), (1, 640, 1), (1, 832, 1), (1, 2050, 2), (1, 4097, 1),
                                       (8, 1200, 1)])
def test_pipelined_attention_kernel(device, monkeypatch, Bn, S2, H2):
    """flash_attn_d64_pipe_kernel (ALG_ATTN_PP=4, the default main launch of the pre-scaled call, and its 4-wave form 3): the asm steady-state loop
    (entered at tile 1 by waves whose running offsets are all zero, whole groups of four tiles) inside its C++ frame, against
    fp32 SDPA and against the straight loop (ALG_ATTN_PP=0) on the same tensors.  Rows whose first-tile max is beyond +-64 keep a
    non-zero offset (those waves never enter the statement while their neighbours in the workgroup do: mixed mode under one
    barrier / DMA protocol); a late dominant key makes a row sum leave [0, 2^80) INSIDE the statement (it bails out, the tile is
    redone on the exact path); ragged tails; S = 512 / 513 stay below the statement's minimum of eight tiles; 8 x 1 heads let
    the split-KV tail plan engage next to the pipelined main launch.  Run-to-run identical."""
    g = torch.Generator().manual_seed(S2 + H2)
    c = 0.125 * 1.4426950408889634
    q, k, v = rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g), rnd((Bn, S2, H2, 64), g)
    q[:, : S2 // 5] *= 9.0                       # scores ~ +-70 in log2 units: non-zero offsets for a fifth of the rows
    k[:, (2 * S2) // 3] *= 6.0                   # one late key that dominates
    D = H2 * 64
    S_pad = (S2 + 127) // 128 * 128
    qs = (q.float() * c).to(BF)
    sc2 = torch.einsum("bqhd,bkhd->bhqk", qs.double(), k.double()) * math.log(2.0)
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc2, dim=-1), v.double())
    qkb = torch.cat([qs.reshape(Bn, S2, D), k.reshape(Bn, S2, D)], dim=-1).contiguous().to(device)
    vt = torch.zeros(Bn, D, S_pad, dtype=BF)
    vt[:, :, torch.tensor([swap23(n) for n in range(S2)])] = v.reshape(Bn, S2, D).transpose(1, 2)
    vt = vt.to(device)
    outs, errs = {}, {}
    # 7 = the 8-wave statement on v_mfma_f32_16x16x32_bf16 (attention64_m16.hip, round 6), 4 = the 8-wave statement on 32x32x16,
    # 0 = the straight loop
    forms = ("7", "4", "0")
    for pp in forms:
        monkeypatch.setenv("ALG_ATTN_PP", pp)
        o = torch.full((Bn, S2, D), 3.0, dtype=BF, device=device)
        _lib.flash_attn_d64(qkb, qkb, vt, o, Bn, H2, S2, S2 * 2 * D, 2 * D, D * S_pad, S_pad, S2 * D, D, 0.125, k_off=D,
                            q_prescaled=True)
        outs[pp] = o
        got = o.cpu().reshape(Bn, S2, H2, 64).double()
        assert torch.isfinite(got).all(), pp
        errs[pp] = ((got - ref).abs().max().item(), (got - ref).abs().mean().item())
    assert errs["0"][0] <= 3e-2 and errs["0"][1] <= 2e-3, errs
    for pp in forms[:-1]:
        assert errs[pp][0] <= 3e-2 and errs[pp][1] <= 2e-3, errs
        assert errs[pp][1] <= 1.25 * errs["0"][1] + 1e-5, errs       # not worse than the straight loop on average
    for pp in ("4", "7"):
        monkeypatch.setenv("ALG_ATTN_PP", pp)
        for _ in range(3):
            o2 = torch.empty_like(outs[pp])
            _lib.flash_attn_d64(qkb, qkb, vt, o2, Bn, H2, S2, S2 * 2 * D, 2 * D, D * S_pad, S_pad, S2 * D, D, 0.125, k_off=D,
                                q_prescaled=True)
            assert torch.equal(o2, outs[pp]), pp
