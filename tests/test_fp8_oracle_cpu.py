"""CPU checks of oracle/fp8_oracle.py (BASELINE config 5's e4m3 operand arithmetic) -- the number format is pinned by the
third-party package itself: a from-the-definition e4m3 rounding against `torch.float8_e4m3fn` on every code and on a dense
sweep; the scheme (per-token / per-channel amax / 448) against hand-computed cases; the `fp8=True` mode of the Wan oracle
routes exactly the seven block linears the product quantises (alg_amd/transformer_wan.py) and nothing else."""
import math

import pytest
import torch

from oracle import fp8_oracle, wan_oracle

F8 = torch.float8_e4m3fn


def test_every_e4m3_code_and_the_format_constants():
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(F8).float()
    finite = vals[torch.isfinite(vals)]
    assert finite.numel() == 254                                    # 0x7f and 0xff are NaN, there are no infinities
    assert finite.max().item() == 448.0 and finite.min().item() == -448.0
    pos = finite[finite > 0]
    assert pos.min().item() == 2.0 ** -9                            # smallest subnormal
    assert torch.equal(fp8_oracle.e4m3_round(finite), finite)       # representable values are fixed points
    # decode from the definition: sign, 4 exponent bits (bias 7), 3 mantissa bits
    c = codes[torch.isfinite(vals)].int()
    s, e, m = (c >> 7) & 1, (c >> 3) & 15, c & 7
    want = torch.where(e == 0, m.float() * 2.0 ** -9, (1 + m.float() / 8) * torch.pow(2.0, (e - 7).float()))
    assert torch.equal(torch.where(s == 1, -want, want), finite)


def test_from_the_definition_rounding_equals_torchs_cast_on_a_dense_sweep():
    g = torch.Generator().manual_seed(0)
    mag = torch.exp(torch.rand(400_000, generator=g) * (math.log(448.0) - math.log(2.0 ** -12)) + math.log(2.0 ** -12))
    x = mag * torch.where(torch.rand(400_000, generator=g) < 0.5, -1.0, 1.0)
    finite = torch.arange(256, dtype=torch.uint8).view(F8).float()
    finite = finite[torch.isfinite(finite)].sort().values
    mid = (finite[1:] + finite[:-1]) / 2                            # every tie: must go to the even mantissa
    x = torch.cat([x, mid, torch.nextafter(mid, torch.tensor(1e9)), torch.nextafter(mid, torch.tensor(-1e9)),
                   torch.tensor([0.0, -0.0, 448.0, -448.0, 1e-30])])
    assert torch.equal(fp8_oracle.e4m3_round(x), x.to(F8).float())
    # values past the range are clamped by the oracle (torch's cast alone would not saturate: the product clamps too)
    assert fp8_oracle.e4m3_round(torch.tensor([1e4, -1e4, 460.0])).tolist() == [448.0, -448.0, 448.0]


def test_row_quantiser_scheme():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(7, 256, generator=g) * 3).to(torch.bfloat16)
    x[0] = 0
    x[1, 5] = 1000.0
    q, s = fp8_oracle.quantize_rows(x)
    q2, s2 = fp8_oracle.quantize_rows(x, via_torch_cast=False)
    assert torch.equal(q, q2) and torch.equal(s, s2)               # the two e4m3 roundings agree
    assert s[0].item() == 1.0 and bool((q[0] == 0).all())          # an all-zero row keeps scale 1
    amax = x.float().abs().amax(1)
    assert torch.allclose(s[1:], amax[1:] / 448.0, rtol=1e-6, atol=0)
    assert q.abs().max().item() == 448.0 and q[1, 5].item() == 448.0    # the row maximum lands on the top code
    assert torch.equal(q.to(F8).float(), q)                        # every value representable
    # relative error of a de-quantised NORMAL value <= 2^-4 (3 mantissa bits), of any value <= half a subnormal step
    d = fp8_oracle.dequantize(q, s)
    err = (d - x.float()).abs()
    assert bool((err <= torch.maximum(x.float().abs() * 2.0 ** -4, s[:, None] * 2.0 ** -10) * (1 + 1e-6)).all())


def test_linear_equals_the_dequantised_matmul_and_is_chunk_invariant():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 50, 384, generator=g).to(torch.bfloat16)
    w = (torch.randn(96, 384, generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(96, generator=g) * 0.1).to(torch.bfloat16)
    y = fp8_oracle.linear(x, w, b)
    assert y.shape == (3, 50, 96) and y.dtype == torch.bfloat16
    qx, sx = fp8_oracle.quantize_rows(x.reshape(-1, 384))
    qw, sw = fp8_oracle.quantize_rows(w)
    want = ((qx.double() @ qw.double().t()) * sx.double()[:, None] * sw.double()[None] + b.double()).float().to(torch.bfloat16)
    assert (y.reshape(-1, 96).float() - want.float()).abs().max().item() <= 2.0 ** -8 * want.float().abs().max().item()
    assert torch.equal(fp8_oracle.linear(x, w, b, chunk_rows=7), y)
    # quantisation error against the unquantised product sits where the format says it should
    full = torch.nn.functional.linear(x.float(), w.float(), b.float())
    r = ((y.float() - full).norm() / full.norm()).item()
    assert 0.3 * fp8_oracle.quantization_error_bound(384) < r < 1.5 * fp8_oracle.quantization_error_bound(384), r


def test_wan_oracle_fp8_mode_routes_exactly_the_seven_block_linears(monkeypatch):
    cfg = wan_oracle.WanConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=64, image_dim=64,
                               added_kv_proj_dim=256)
    sd = wan_oracle.init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 36, 2, 8, 8, generator=g)
    txt, img = torch.randn(2, 16, 64, generator=g), torch.randn(2, 9, 64, generator=g)
    t = torch.tensor([500.0, 500.0])
    seen = []
    real = fp8_oracle.linear

    def spy(xx, w, b, **kw):
        seen.append(tuple(w.shape))
        return real(xx, w, b, **kw)

    monkeypatch.setattr(fp8_oracle, "linear", spy)
    ref = wan_oracle.wan_forward(cfg, sd, x, t, txt, img)
    assert not seen
    out = wan_oracle.wan_forward(cfg, sd, x, t, txt, img, fp8=True)
    D, Ff = cfg.dim, cfg.ffn_dim
    per_block = [(D, D)] * 4 + [(D, D)] * 2 + [(Ff, D), (D, Ff)]          # q, k, v, out, cross q, cross out, ff1, ff2
    assert sorted(seen) == sorted(per_block * cfg.num_layers)
    r = ((out - ref).norm() / ref.norm()).item()
    assert 2e-3 < r < 0.15, r                                             # a real, bounded quantisation effect
    # bf16-eager fp8 mode (the floor the GPU tests anchor to) runs and stays close to the fp32 fp8 mode
    eager = wan_oracle.wan_forward(cfg, sd, x.bfloat16(), t, txt.bfloat16(), img.bfloat16(), dtype=torch.bfloat16, fp8=True)
    assert eager.dtype == torch.bfloat16
    assert ((eager.float() - out).norm() / out.norm()).item() < 0.1


def test_chunked_sdpa_is_the_same_arithmetic():
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(2, n, 256, generator=g) for n in (70, 90, 90))
    a = wan_oracle._sdpa(q, k, v, 2)
    b = wan_oracle._sdpa(q, k, v, 2, max_scores=90 * 16)                 # 16-row pieces
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
