"""GPU parity of the Wan-DiT building blocks against plain torch fp32 on the device."""
import math

import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def _perm(n):
    return torch.tensor([(i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1) for i in range(n)], device=DEV)


def make_vt(v, s_pad):
    """v [B, S, H*128] -> V^T [B, H*128, s_pad] with kv index bits 2 and 3 swapped, zero padded."""
    B, S, D = v.shape
    vt = torch.zeros(B, D, s_pad, dtype=BF, device=DEV)
    vt[:, :, _perm(s_pad)[:S]] = v.transpose(1, 2)  # logical s sits at position perm(s)
    return vt


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 300, 300), (2, 3, 777, 257), (1, 2, 520, 512), (1, 1, 31, 64),
                                          (1, 2, 1024, 2050)])
def test_flash_attn_d128_matches_sdpa(B, H, Sq, Skv):
    D = H * 128
    q, k, v = _rand((B, Sq, D), 1), _rand((B, Skv, D), 2), _rand((B, Skv, D), 3)
    s_pad = (Skv + 63) // 64 * 64
    vt = make_vt(v, s_pad)
    o = torch.zeros(B, Sq, D, dtype=BF, device=DEV)
    scale = 1.0 / math.sqrt(128)
    _lib.flash_attn_d128(q, k, vt, o, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
    qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
    kh = k.float().view(B, Skv, H, 128).transpose(1, 2)
    vh = v.float().view(B, Skv, H, 128).transpose(1, 2)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh
    ref = ref.transpose(1, 2).reshape(B, Sq, D)
    err = (o.float() - ref).abs().max().item()
    assert err < 2e-2, err  # bf16 P and bf16 output on O(1) values
    assert (o.float() - ref).abs().mean().item() < 2e-3


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 700, 1024), (2, 3, 257, 1000), (1, 1, 256, 513), (1, 2, 1300, 2050),
                                          (1, 1, 64, 544), (1, 2, 520, 575), (1, 1, 300, 640), (1, 2, 100, 700),
                                          (1, 1, 200, 832)])
def test_flash_attn_d128_q64_kernel(B, H, Sq, Skv, monkeypatch):
    """The long self-attention form (attention128_q64.hip: 64 queries per wave, software-pipelined 32-key half-tiles; the default
    from 4,096 keys on, here forced onto every call of >= 8 KV tiles with ALG_ATTN128_Q64=2): against fp32 SDPA and against the
    32-query kernels (ALG_ATTN128_Q64=0) on the same tensors.  Ragged key
    counts: a last tile whose second half is partly (1000, 2050), entirely (513, 544: 1 / 32 keys) masked, or exactly half
    (575 = 8 x 64 + 63), query blocks that end mid-wave, scores large enough to trip the lazy running max's exact path; tile
    counts that leave 0, 1, 2 and 3 tiles to the runtime-slot remainder of the four-tile unrolled main loop."""
    D = H * 128
    q, k, v = _rand((B, Sq, D), 11), _rand((B, Skv, D), 12), _rand((B, Skv, D), 13)
    q[:, : Sq // 2] *= 6.0                       # half the queries: scores ~ +-25 -> row sums far beyond the first tile's
    k[:, Skv // 3] *= 8.0                        # and one key that dominates late
    s_pad = (Skv + 63) // 64 * 64
    vt = make_vt(v, s_pad)
    scale = 1.0 / math.sqrt(128)
    outs = {}
    for flag in ("2", "3", "0"):       # 2: statement + frame, 3: the frame's C++ tile body on its own, 0: the 32-query kernels
        monkeypatch.setenv("ALG_ATTN128_Q64", flag)
        o = torch.full((B, Sq, D), 7.0, dtype=BF, device=DEV)
        _lib.flash_attn_d128(q, k, vt, o, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
        outs[flag] = o
    qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
    kh = k.float().view(B, Skv, H, 128).transpose(1, 2)
    vh = v.float().view(B, Skv, H, 128).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    for flag, o in outs.items():
        assert bool(torch.isfinite(o.float()).all()), flag
        assert (o.float() - ref).abs().max().item() < 3e-2, flag
        assert (o.float() - ref).abs().mean().item() < 2e-3, flag
    # same arithmetic per query (S^T = K Q^T, bf16 P, fp32 O); only the lazy max's offsets may differ (32- vs 64-key steps)
    assert (outs["2"].float() - outs["0"].float()).abs().max().item() < 1.6e-2
    monkeypatch.setenv("ALG_ATTN128_Q64", "2")
    o2 = torch.empty_like(outs["2"])
    _lib.flash_attn_d128(q, k, vt, o2, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
    assert torch.equal(o2, outs["2"])             # deterministic


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 700, 1024), (2, 3, 257, 1000), (1, 1, 256, 833), (1, 2, 1300, 2050),
                                          (1, 1, 64, 1100), (1, 2, 520, 767), (1, 1, 300, 768), (1, 2, 100, 1279),
                                          (1, 1, 200, 4097)])
def test_flash_attn_d128_pipelined_kernel(B, H, Sq, Skv, monkeypatch):
    """attention128_pipe.hip (ALG_ATTN128_PIPE=1: generated asm steady-state loop -- PV(t-1) / QK(t+1) / softmax(t) pipelined, every
    MFMA followed in the same wave by a slice of the softmax and a fragment read -- inside a C++ frame, taken for >= 12 KV tiles of
    non-causal, ungrouped attention) against fp32 SDPA and against attention128.hip on the same tensors.  Ragged key counts (the
    statement never sees the masked tile), scores large enough for the lazy running max's exact path INSIDE the statement's
    range (it bails out, the tile is redone in C++), a late dominant key, query blocks that end mid-wave, tile counts that
    leave 0-3 tiles behind the four-tile groups; run-to-run identical."""
    D = H * 128
    q, k, v = _rand((B, Sq, D), 11), _rand((B, Skv, D), 12), _rand((B, Skv, D), 13)
    q[:, : Sq // 2] *= 6.0                       # half the queries: scores ~ +-25 -> row sums far beyond the first tile's
    k[:, (2 * Skv) // 3] *= 8.0                  # and one key that dominates late
    s_pad = (Skv + 63) // 64 * 64
    vt = make_vt(v, s_pad)
    scale = 1.0 / math.sqrt(128)
    outs = {}
    monkeypatch.setenv("ALG_ATTN128_Q64", "0")   # (4,097 keys would go to the 64-query kernel by default)
    for flag in ("1", "0"):
        monkeypatch.setenv("ALG_ATTN128_PIPE", flag)
        o = torch.full((B, Sq, D), 7.0, dtype=BF, device=DEV)
        _lib.flash_attn_d128(q, k, vt, o, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
        outs[flag] = o
    qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
    kh = k.float().view(B, Skv, H, 128).transpose(1, 2)
    vh = v.float().view(B, Skv, H, 128).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    for flag, o in outs.items():
        assert bool(torch.isfinite(o.float()).all()), flag
        assert (o.float() - ref).abs().max().item() < 3e-2, flag
        assert (o.float() - ref).abs().mean().item() < 2e-3, flag
    assert (outs["1"].float() - outs["0"].float()).abs().max().item() < 1.6e-2
    monkeypatch.setenv("ALG_ATTN128_PIPE", "1")
    for _ in range(3):
        o2 = torch.empty_like(outs["1"])
        _lib.flash_attn_d128(q, k, vt, o2, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
        assert torch.equal(o2, outs["1"])             # deterministic


@pytest.mark.parametrize("B,H,Sq,Skv,Skv2", [(1, 2, 300, 512, 257), (2, 3, 777, 257, 512), (1, 1, 31, 64, 1), (1, 2, 1024, 130, 70),
                                               (2, 1, 257, 1, 513)])
def test_flash_attn_d128_dual_is_two_launches_and_the_add(B, H, Sq, Skv, Skv2):
    """alg_flash_attn_d128_dual (the text + image cross-attention of the Wan I2V block as one launch) against what it replaces --
    two alg_flash_attn_d128 launches and alg_lincomb -- bit for bit, and against softmax(QK^T)V + softmax(QK2^T)V2 in fp32."""
    D = H * 128
    q, k, v = _rand((B, Sq, D), 1), _rand((B, Skv, D), 2), _rand((B, Skv, D), 3)
    k2, v2 = _rand((B, Skv2, D), 4, 1.5), _rand((B, Skv2, D), 5)
    pad, pad2 = (Skv + 63) // 64 * 64, (Skv2 + 127) // 128 * 128          # (two different paddings: the strides are independent)
    vt, vt2 = make_vt(v, pad), make_vt(v2, pad2)
    scale = 1.0 / math.sqrt(128)
    o_a = torch.zeros(B, Sq, D, dtype=BF, device=DEV)
    o_b = torch.zeros(B, Sq, D, dtype=BF, device=DEV)
    _lib.flash_attn_d128(q, k, vt, o_a, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * pad, pad, Sq * D, D, scale)
    _lib.flash_attn_d128(q, k2, vt2, o_b, B, H, Sq, Skv2, Sq * D, D, Skv2 * D, D, D * pad2, pad2, Sq * D, D, scale)
    two = torch.empty_like(o_a)
    _lib.lincomb([(1.0, o_a), (1.0, o_b)], BF, out=two)
    assert torch.equal(two, o_a + o_b)                                      # (the add kernel is the eager bf16 add)
    one = torch.full((B, Sq, D), 7.0, dtype=BF, device=DEV)
    _lib.flash_attn_d128_dual(q, k, vt, Skv, Skv * D, D, D * pad, pad, k2, vt2, Skv2, Skv2 * D, D, D * pad2, pad2, one, B, H, Sq,
                              Sq * D, D, Sq * D, D, scale)
    assert torch.equal(one, two)

    def sdpa(kk, vv, n):
        qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
        kh = kk.float().view(B, n, H, 128).transpose(1, 2)
        vh = vv.float().view(B, n, H, 128).transpose(1, 2)
        return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    ref = sdpa(k, v, Skv) + sdpa(k2, v2, Skv2)
    assert (one.float() - ref).abs().max().item() < 4e-2
    assert (one.float() - ref).abs().mean().item() < 4e-3


def test_flash_attn_d128_dual_rejects_bad_arguments():
    q = _rand((1, 64, 128), 4)
    ok = dict(k=q, vt=q, n=64, k2=q, vt2=q, n2=64)
    call = lambda **kw: _lib.flash_attn_d128_dual(
        q, kw.get("k", q), kw.get("vt", q), kw.get("n", 64), 64 * 128, 128, 128 * 64, 64, kw.get("k2", q), kw.get("vt2", q),
        kw.get("n2", 64), 64 * 128, 128, 128 * 64, 64, q.clone(), 1, 1, 64, 64 * 128, 128, 64 * 128, 128, 0.1)
    call(**ok)
    with pytest.raises(_lib.AlgHipError):      # second set's vt row stride shorter than Skv2 rounded up
        call(n2=100)
    with pytest.raises(_lib.AlgHipError):
        call(n=0)
    with pytest.raises(_lib.AlgHipError):
        call(k2=q.cpu())


def test_flash_attn_d128_rejects_bad_arguments():
    q = _rand((1, 64, 128), 4)
    with pytest.raises(_lib.AlgHipError):  # vt row stride shorter than Skv rounded up
        _lib.flash_attn_d128(q, q, q, q.clone(), 1, 1, 64, 100, 64 * 128, 128, 64 * 128, 128, 128 * 64, 64, 64 * 128, 128,
                             0.1)
    with pytest.raises(_lib.AlgHipError):
        _lib.flash_attn_d128(q.cpu(), q, q, q.clone(), 1, 1, 64, 64, 64 * 128, 128, 64 * 128, 128, 128 * 64, 64,
                             64 * 128, 128, 0.1)


@pytest.mark.parametrize("D,affine,mod", [(512, False, True), (5120, True, False), (1280, True, False), (1024, True, True)])
def test_layernorm_mod_f32(D, affine, mod):
    B, S = 2, 37
    x = _rand((B, S, D), 5)
    g = torch.Generator().manual_seed(6)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    b = (0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    modv = torch.randn(B, 6, D, generator=g).to(DEV) * 0.3
    y = torch.empty_like(x)
    _lib.layernorm_mod_f32(x, y, w, b, modv if mod else None, modv if mod else None, 6 * D, B, S, D, 1e-6,
                           scale_off=D, shift_off=2 * D)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-6)
    if mod:
        ref = ref * (1 + modv[:, 1:2]) + modv[:, 2:3]
    ref = ref.to(BF)
    assert (y.float() - ref.float()).abs().max().item() <= 2.0 ** -6          # one bf16 ulp at |y| < 4
    assert (y != ref).float().mean().item() < 0.01


@pytest.mark.parametrize("D,affine,mod", [(512, False, True), (5120, True, False), (1536, False, True), (1024, True, True)])
def test_layernorm_mod_f32_fp8_is_norm_then_quantiser(D, affine, mod):
    """The fused fp8 output is bit for bit alg_layernorm_mod_f32 followed by alg_quantize_fp8_rows (bytes and scales)."""
    B, S = 2, 37
    x = _rand((B, S, D), 15)
    x[0, 3] = 0                                              # an all-zero row: scale 1, zero bytes (or the shift alone)
    g = torch.Generator().manual_seed(16)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    b = (0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    modv = (torch.randn(B, 6, D, generator=g) * 0.3).to(DEV) if mod else None
    y = torch.empty_like(x)
    _lib.layernorm_mod_f32(x, y, w, b, modv, modv, 6 * D, B, S, D, 1e-6, scale_off=D, shift_off=2 * D)
    q_want = torch.empty(B * S, D, dtype=torch.uint8, device=DEV)
    s_want = torch.empty(B * S, dtype=torch.float32, device=DEV)
    _lib.quantize_fp8_rows(y, q_want, s_want, B * S, D)
    q_got = torch.full_like(q_want, 0x55)
    s_got = torch.full_like(s_want, -1.0)
    _lib.layernorm_mod_f32_fp8(x, q_got, s_got, w, b, modv, modv, 6 * D, B, S, D, 1e-6, scale_off=D, shift_off=2 * D)
    assert torch.equal(s_got, s_want)
    assert torch.equal(q_got, q_want)


@pytest.mark.parametrize("D,rows,batch", [(5120, 5003, 2), (1536, 4100, 1), (3072, 2100, 3)])
def test_layernorm_mod_f32_many_rows_form_is_bit_identical(D, rows, batch):
    """From 4,096 rows on the modulated non-affine LayerNorm (norm1 / norm3 of the Wan blocks) runs with its fp32 parameters in the LDS
    and several rows per wave (ln_mod_f32_rows_kernel); the same rows through calls of fewer than 4,096 rows take the one-row-per-wave
    kernel: the bf16 output and the e4m3 bytes + row scales must be the same bits."""
    x = _rand((batch, rows, D), 31, 2.0)
    g = torch.Generator().manual_seed(32)
    mod = (torch.randn(batch, 2, D, generator=g) * 0.3).float().to(DEV)        # [b][scale | shift][D], fp32
    y = torch.full((batch, rows, D), 5.0, dtype=BF, device=DEV)
    _lib.layernorm_mod_f32(x, y, None, None, mod, mod, 2 * D, batch, rows, D, 1e-6, scale_off=0, shift_off=D)
    q8 = torch.zeros(batch, rows, D, dtype=torch.uint8, device=DEV)
    qs = torch.zeros(batch, rows, dtype=torch.float32, device=DEV)
    _lib.layernorm_mod_f32_fp8(x, q8, qs, None, None, mod, mod, 2 * D, batch, rows, D, 1e-6, scale_off=0, shift_off=D)
    y_ref, q8_ref, qs_ref = torch.zeros_like(y), torch.zeros_like(q8), torch.zeros_like(qs)
    for b in range(batch):
        for a in range(0, rows, 1900):
            n = min(1900, rows - a)
            _lib.layernorm_mod_f32(x[b, a:a + n], y_ref[b, a:a + n], None, None, mod[b, 0], mod[b, 1], 0, 1, n, D, 1e-6)
            _lib.layernorm_mod_f32_fp8(x[b, a:a + n], q8_ref[b, a:a + n], qs_ref[b, a:a + n], None, None, mod[b, 0], mod[b, 1], 0, 1,
                                       n, D, 1e-6)
    assert torch.equal(y, y_ref)
    assert torch.equal(q8, q8_ref) and torch.equal(qs, qs_ref)
    xf = x.float()
    ref = torch.nn.functional.layer_norm(xf, (D,), eps=1e-6) * (1 + mod[:, :1]) + mod[:, 1:]
    assert (y.float() - ref).abs().max().item() < 0.06
    # the affine, unmodulated form (norm2 of a Wan block: fp32 weight and bias shared by the batch) takes the same kernel
    w, b = (1.0 + mod[0, 0]).contiguous(), mod[0, 1].contiguous()
    _lib.layernorm_mod_f32(x, y, w, b, None, None, 0, batch, rows, D, 1e-6)
    for bi in range(batch):
        for a in range(0, rows, 1900):
            n = min(1900, rows - a)
            _lib.layernorm_mod_f32(x[bi, a:a + n], y_ref[bi, a:a + n], w, b, None, None, 0, 1, n, D, 1e-6)
    assert torch.equal(y, y_ref)


def test_layernorm_mod_f32_fp8_rejects_unsupported_width():
    x = _rand((1, 4, 1280), 17)
    q = torch.empty(4, 1280, dtype=torch.uint8, device=DEV)
    s = torch.empty(4, dtype=torch.float32, device=DEV)
    with pytest.raises(_lib.AlgHipError, match="512"):
        _lib.layernorm_mod_f32_fp8(x, q, s, None, None, None, None, 0, 1, 4, 1280, 1e-6)
    with pytest.raises(_lib.AlgHipError, match="null"):
        _lib.layernorm_mod_f32_fp8(x, q, None, None, None, None, None, 0, 1, 4, 1024, 1e-6)


@pytest.mark.parametrize("D,rope", [(512, True), (5120, True), (512, False)])
def test_rmsnorm_rope(D, rope):
    B, S = 2, 29
    x = _rand((B, S, 2 * D), 7)
    w = _rand((D,), 8, 0.1) + 1
    g = torch.Generator().manual_seed(9)
    ang = torch.rand(S, 64, generator=g, dtype=torch.float64) * 6.0
    cos, sin = torch.cos(ang), torch.sin(ang)
    y = x.clone()
    _lib.rmsnorm_rope_(y, w, cos.float().to(DEV) if rope else None, sin.float().to(DEV) if rope else None, 2 * D, B, S, D,
                       1e-6, x_off=D)  # the k half of a fused [q | k] buffer
    xs = x[:, :, D:]
    var = xs.float().pow(2).mean(-1, keepdim=True)
    ref = (xs.float() * torch.rsqrt(var + 1e-6)).to(BF) * w
    if rope:
        r = ref.to(torch.float64).view(B, S, D // 128, 64, 2)
        c, s_ = cos.to(DEV)[None, :, None, :], sin.to(DEV)[None, :, None, :]
        ref = torch.stack([r[..., 0] * c - r[..., 1] * s_, r[..., 0] * s_ + r[..., 1] * c], dim=-1).reshape(B, S, D).to(BF)
    assert torch.equal(y[:, :, :D], x[:, :, :D])          # the q half is untouched
    got = y[:, :, D:]
    assert (got.float() - ref.float()).abs().max().item() <= 2.0 ** -5
    assert (got != ref).float().mean().item() < 0.02


def test_patchify_unpatchify_modulation_linear():
    N, C, F, H, W = 2, 36, 3, 8, 12
    x = _rand((N, C, F, H, W), 10)
    S = F * (H // 2) * (W // 2)
    out = torch.empty(N, S, 192, dtype=BF, device=DEV)
    _lib.patchify3d(x, out, N, C, F, H, W, 2, 2, 192)
    ref = x.view(N, C, F, H // 2, 2, W // 2, 2).permute(0, 2, 3, 5, 1, 4, 6).reshape(N, S, 144)
    assert torch.equal(out[:, :, :144], ref) and not out[:, :, 144:].any()
    tok = _rand((N, S, 64), 11)
    img = torch.empty(N, 16, F, H, W, dtype=BF, device=DEV)
    _lib.unpatchify3d(tok, 64, img, N, 16, F, H, W, 2, 2)
    ref = tok.reshape(N, F, H // 2, W // 2, 1, 2, 2, 16).permute(0, 7, 1, 4, 2, 5, 3, 6).flatten(6, 7).flatten(4, 5).flatten(2, 3)
    assert torch.equal(img, ref)
    g = torch.Generator().manual_seed(12)
    table = torch.randn(3, 6, 512, generator=g).to(DEV)
    vec = _rand((N, 6 * 512), 13)
    mod = torch.empty(3, N, 6, 512, device=DEV)
    _lib.wan_modulation(table, vec, mod, 3, N, 6, 512, True)
    assert torch.equal(mod, table[:, None] + vec.float().view(1, N, 6, 512))
    t = torch.tensor([999.0, 12.0], device=DEV)
    emb = torch.empty(2, 256, device=DEV)
    _lib.timestep_embedding_f32(t, emb, 2, 256)
    freq = torch.exp(-math.log(10000.0) * torch.arange(128, dtype=torch.float32, device=DEV) / 128)
    a = t[:, None] * freq[None]
    assert torch.allclose(emb, torch.cat([torch.cos(a), torch.sin(a)], dim=-1), atol=2e-4)
    Wt, bt = torch.randn(64, 256, generator=g).to(DEV) / 16, torch.randn(64, generator=g).to(DEV)
    y, ybf, ysl = torch.empty(2, 64, device=DEV), torch.empty(2, 64, dtype=BF, device=DEV), torch.empty(2, 64, dtype=BF, device=DEV)
    _lib.linear_f32(emb, Wt, bt, y, ybf, ysl, 2, 64, 256, act=0)
    ref = emb @ Wt.t() + bt
    assert torch.allclose(y, ref, atol=1e-5)
    assert torch.equal(ybf, y.to(BF))
    assert (ysl.float() - torch.nn.functional.silu(y.to(BF)).float()).abs().max().item() <= 2.0 ** -7
    z = _rand((5, 77), 14)
    ref = torch.nn.functional.gelu(z.float()).to(BF)
    _lib.gelu_erf_(z)
    assert (z.float() - ref.float()).abs().max().item() <= 2.0 ** -7


def test_gemm_fp32_gate_epilogue():
    B, S, D, K = 2, 300, 512, 256
    a, wgt, bias, r = _rand((B, S, K), 15), _rand((D, K), 16, 0.06), _rand((D,), 17, 0.1), _rand((B, S, D), 18)
    g = torch.Generator().manual_seed(19)
    gate = torch.randn(B, 6, D, generator=g).to(DEV)
    c = r.clone()
    _lib.gemm(a, wgt, c, S, D, K, K, K, D, bias=bias, R=c, ldr=D, gate=gate, gate_off=2 * D, strideGate=6 * D, batch=B,
              strideA=S * K, strideC=S * D, strideR=S * D, seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)
    lin = (a.float() @ wgt.float().t() + bias.float()).to(BF)
    ref = (r.float() + lin.float() * gate[:, 2:3]).to(BF)
    assert (c.float() - ref.float()).abs().max().item() <= 2.0 ** -5
    assert (c != ref).float().mean().item() < 0.02
