"""GPU parity of the fp8 (OCP e4m3) operand path of BASELINE config 5: the row quantiser against torch's float8_e4m3fn
cast, the fp8 GEMM against an fp32 matmul of the de-quantised operands (products of e4m3 values are exact in fp32)."""
import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def quant(x):
    rows, K = x.shape
    q = torch.empty(rows, K, dtype=torch.uint8, device=DEV)
    s = torch.empty(rows, dtype=torch.float32, device=DEV)
    _lib.quantize_fp8_rows(x, q, s, rows, K)
    return q, s


@pytest.mark.parametrize("rows,K", [(5, 512), (300, 5120), (7, 1152), (41, 13824), (9, 3072)])   # (5120, 13824, 3072: the row-in-registers form)
def test_quantize_fp8_rows_matches_torch(rows, K):
    x = _rand((rows, K), 1, 3.0)
    x[0] = 0                                           # an all-zero row keeps scale 1
    q, s = quant(x)
    amax = x.float().abs().amax(dim=1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s, want_s, rtol=1e-6)
    want_q = (x.float() * (1.0 / want_s)[:, None]).clamp(-448, 448).to(F8)
    assert torch.equal(q.view(F8).float(), want_q.float())


def test_quantize_fp8_rows_strided_input_is_the_contiguous_result():
    rows, K, pitch = 37, 5120, 5120 + 13824
    big = _rand((rows, pitch), 6, 2.0)
    q0, s0 = quant(big[:, 13824:].contiguous())
    q1 = torch.empty(rows, K, dtype=torch.uint8, device=DEV)
    s1 = torch.empty(rows, dtype=torch.float32, device=DEV)
    _lib.quantize_fp8_rows(big, q1, s1, rows, K, x_rstride=pitch, x_off=13824)
    assert torch.equal(q0, q1) and torch.equal(s0, s1)


@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (1000, 640, 1152), (33, 64, 128)])
def test_gemm_fp8_matches_dequantised_matmul(M, N, K):
    a, w, bias = _rand((M, K), 2), _rand((N, K), 3, 0.05), _rand((N,), 4, 0.1)
    qa, sa = quant(a)
    qw, sw = quant(w)
    c = torch.empty(M, N, dtype=BF, device=DEV)
    _lib.gemm(qa, qw, c, M, N, K, K, K, N, bias=bias, a_scale=sa, b_scale=sw)
    ref = (qa.view(F8).float() * sa[:, None]) @ (qw.view(F8).float() * sw[:, None]).t() + bias.float()
    ref = ref.to(BF)
    assert (c.float() - ref.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item())
    assert (c != ref).float().mean().item() < 0.02
    # and the quantisation itself stays within the e4m3 error budget of the bf16 product
    full = (a.float() @ w.float().t() + bias.float())
    assert ((c.float() - full).norm() / full.norm()).item() < 0.06


def test_gemm_fp8_epilogues():
    """Residual + fp32 gate (Wan), GELU, transposed / permuted V^T store -- the epilogues the Wan DiT uses."""
    B, S, D, K = 2, 200, 512, 256
    a, w, bias = _rand((B, S, K), 5), _rand((D, K), 6, 0.06), _rand((D,), 7, 0.1)
    qa, sa = quant(a.view(B * S, K))
    qw, sw = quant(w)
    deq = lambda q, s: q.view(F8).float() * s[:, None]
    lin = (deq(qa, sa) @ deq(qw, sw).t() + bias.float()).to(BF).view(B, S, D)
    r = _rand((B, S, D), 8)
    g = torch.Generator().manual_seed(9)
    gate = torch.randn(B, 6, D, generator=g).to(DEV)
    c = r.clone()
    _lib.gemm(qa, qw, c, S, D, K, K, K, D, bias=bias, R=c, ldr=D, gate=gate, gate_off=2 * D, strideGate=6 * D, batch=B,
              strideA=S * K, strideC=S * D, strideR=S * D, seg_split=1 << 30, flags=_lib.GEMM_GATE_F32, a_scale=sa,
              b_scale=sw, strideAScale=S)
    ref = (r.float() + lin.float() * gate[:, 2:3]).to(BF)
    assert (c.float() - ref.float()).abs().max().item() <= 2.0 ** -5
    h = torch.empty(B * S, D, dtype=BF, device=DEV)
    _lib.gemm(qa, qw, h, B * S, D, K, K, K, D, bias=bias, act=_lib.ACT_GELU_TANH, a_scale=sa, b_scale=sw)
    ref = torch.nn.functional.gelu(lin.float(), approximate="tanh").to(BF).view(B * S, D)
    assert (h.float() - ref.float()).abs().max().item() <= 2.0 ** -5
    # V^T: weights as the A operand (rows = channels), tokens as B; bias per row; kv index bits 2 and 3 swapped
    s_pad = 256
    vt = torch.zeros(B, D, s_pad, dtype=BF, device=DEV)
    _lib.gemm(qw, qa, vt, D, S, K, K, K, s_pad, bias=bias, batch=B, strideB=S * K, strideC=D * s_pad,
              flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS, a_scale=sw, b_scale=sa, strideBScale=S)
    perm = torch.tensor([(i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1) for i in range(s_pad)], device=DEV)
    got = vt[:, :, perm[:S]].transpose(1, 2)
    assert (got.float() - lin.float()).abs().max().item() <= 2.0 ** -5


def test_tall_operand_is_cut_into_slabs_with_every_per_row_operand_following():
    """M * ldc >= 2^31 (CFG samples of a 75,600- / 118,800-token sequence flattened into M): alg_gemm_* validates the whole
    call, then launches tile-aligned slabs.  Per-row operands (A, C, R in place, the fp8 row scales, the segment split of the
    fp32 gate) must move with the slab: the tall call equals two independent calls on row ranges that stay below the limit."""
    M, N, K = 600_000, 4096, 128
    split = 560_000                                     # inside the second slab (slabs are 524,032 rows at ldc = 4096)
    assert M * N >= 2 ** 31
    g = torch.Generator(device=DEV).manual_seed(21)
    a = torch.randn(M, K, generator=g, device=DEV).to(BF)
    w, bias = _rand((N, K), 22, 0.06), _rand((N,), 23, 0.1)
    qa, sa = quant(a)
    qw, sw = quant(w)
    del a
    gate = torch.randn(1, 2, N, generator=g, device=DEV)
    r = torch.randn(M, N, generator=g, device=DEV, dtype=BF)
    kw = dict(bias=bias, ldr=N, gate=gate, flags=_lib.GEMM_GATE_F32, b_scale=sw)
    tall = r.clone()
    _lib.gemm(qa, qw, tall, M, N, K, K, K, N, R=tall, seg_split=split, a_scale=sa, **kw)
    want = r.clone()
    del r
    h = 300_000
    _lib.gemm(qa, qw, want, h, N, K, K, K, N, R=want, seg_split=split, a_scale=sa, **kw)
    _lib.gemm(qa, qw, want, M - h, N, K, K, K, N, R=want, seg_split=split - h, a_scale=sa, a_off=h * K, c_off=h * N,
              r_off=h * N, a_scale_off=h, **kw)
    assert torch.equal(tall, want)
    # the two gate segments really differ across the split, and rows on both sides of a slab boundary were written
    rows = torch.tensor([0, 524_031, 524_032, split - 1, split, M - 1], device=DEV)
    lin = ((qa[rows].view(F8).float() * sa[rows, None]) @ (qw.view(F8).float() * sw[:, None]).t() + bias.float()).to(BF)
    seg = (rows >= split).long()
    assert torch.isfinite(tall[rows].float()).all() and not torch.equal(gate[0, 0], gate[0, 1])
    assert lin.shape == (6, N) and seg.tolist() == [0, 0, 0, 0, 1, 1]


def test_bad_arguments_of_a_tall_call_are_refused_before_any_slab_runs():
    """A mis-aligned residual pitch on a call that needs slabs: refused up front, C untouched."""
    M, N, K = 600_000, 4096, 128
    qa = torch.zeros(M, K, dtype=torch.uint8, device=DEV)
    qw = torch.zeros(N, K, dtype=torch.uint8, device=DEV)
    sa, sw = torch.ones(M, device=DEV), torch.ones(N, device=DEV)
    c = torch.full((M, N), 3.0, dtype=BF, device=DEV)
    with pytest.raises(_lib.AlgHipError):
        _lib.gemm(qa, qw, c, M, N, K, K, K, N, R=c, ldr=N + 2, a_scale=sa, b_scale=sw)
    assert bool((c[:1024] == 3.0).all()) and bool((c[-1024:] == 3.0).all())


@pytest.mark.parametrize("form", ["plain", "gelu", "vt", "res", "res_gate_f32", "res_gate_f32_straddle"])
def test_fp8_schedule9_equals_the_ping_pong_schedule_bit_for_bit(monkeypatch, form):
    """Round 4: e4m3 operands on schedule 9 (generated asm K loop on v_mfma_scale_f32_32x32x64_f8f6f4, gemm_p9_fp8_loop.inc) against
    schedule 6 (compiler-scheduled ping-pong, the same instruction): both feed an MFMA the fragments of k-steps (2 kp, 2 kp + 1) of
    a 128-byte k-tile, k-tiles in order -- the same contraction order, so equal bits.  K / 128 = 2 ... 12 runs the residual
    catch-up chain from each of its labels and past it (K = 128 stays on schedule 6: the loop needs two k-tiles); shapes with edge
    tiles in M and N; every epilogue the Wan DiT uses."""
    g = torch.Generator(device=DEV).manual_seed(9)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device=DEV) * sc).to(BF)
    for M, N, K in [(300, 520, 128 * k) for k in range(1, 13)] + [(1111, 96, 5120), (2100, 1024, 128 * 11)]:
        a, w, bias, x0 = rn(M, K), rn(N, K, sc=0.05), rn(N), rn(M, N)
        qa, sa = quant(a)
        qw, sw = quant(w)
        gate32, brow = torch.randn(1, 2 * N, generator=g, device=DEV), rn(M)

        def run():
            if form == "plain":
                c = torch.full((M, N), 7.0, dtype=BF, device=DEV)
                _lib.gemm(qa, qw, c, M, N, K, K, K, N, bias=bias, a_scale=sa, b_scale=sw)
            elif form == "gelu":
                c = torch.full((M, N), 7.0, dtype=BF, device=DEV)
                _lib.gemm(qa, qw, c, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH, a_scale=sa, b_scale=sw)
            elif form == "vt":      # transposed V projection: A = weights, per-row bias, permuted columns
                n_pad = (M + 63) // 64 * 64
                c = torch.zeros(N, n_pad, dtype=BF, device=DEV)
                _lib.gemm(qw, qa, c, N, M, K, K, K, n_pad, bias=bias, a_scale=sw, b_scale=sa,
                          flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            else:
                c = x0.clone()
                kw = dict(bias=bias, R=c, ldr=N, a_scale=sa, b_scale=sw)
                if form == "res_gate_f32":
                    kw.update(gate=gate32, strideGate=2 * N, seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)
                elif form == "res_gate_f32_straddle":
                    kw.update(gate=gate32, strideGate=2 * N, seg_split=M // 3, flags=_lib.GEMM_GATE_F32)
                _lib.gemm(qa, qw, c, M, N, K, K, K, N, **kw)
            return c

        monkeypatch.setenv("ALG_GEMM_PIPE", "6")
        want = run()
        monkeypatch.setenv("ALG_GEMM_PIPE", "9")
        for _ in range(2):
            got = run()
            if not torch.equal(got, want):
                d = (got.float() - want.float()).abs()
                raise AssertionError("%s M=%d N=%d K=%d: %d elements differ, max %.4g" % (form, M, N, K, int((d > 0).sum()), d.max().item()))


@pytest.mark.parametrize("K,what", [(5120, "out-projection / Q|K / cross-attention contraction"), (13824, "ff2 contraction (Wan ffn_dim)")])
def test_schedule9_fp8_gemm_at_the_c5_shape_against_a_float64_dequantised_matmul(K, what):
    """the schedule-9 fp8 loop (the default) had only met an independent reference at K <= 1152; at the C5
    shapes it was checked by bit-identity with schedule 6.  Here: M = 75,600 tokens (the C5 sequence), N = 5120, K = 5120 and
    K = 13,824 (`ffn_dim`: 108 k-tiles of 128), the DEFAULT schedule, against the de-quantised operands multiplied in float64
    on the host for 96 sampled rows (first / last tile rows, tile borders, random rows).  Bound per element: one bf16 rounding
    of the result (half an ulp: up to 2^-8 relative) + the fp32 accumulation slack 2^-16 * sum_k |a_k w_k| (K fp32 additions of
    exact products, each within 2^-24 of its partial sum: ~ sqrt(K) 2^-24 <= 2^-17 typical, K 2^-24 = 2^-10 worst case)."""
    M, N = 75_600, 5120
    g = torch.Generator(device=DEV).manual_seed(100 + K)
    a = (torch.randn(M, K, generator=g, device=DEV) * torch.rand(M, 1, generator=g, device=DEV).mul(4).exp()).to(BF)   # row scales over e^4
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.03).to(BF)
    bias = (torch.randn(N, generator=g, device=DEV) * 0.1).to(BF)
    qa, sa = quant(a)
    qw, sw = quant(w)
    del a, w
    c = torch.empty(M, N, dtype=BF, device=DEV)
    _lib.gemm(qa, qw, c, M, N, K, K, K, N, bias=bias, a_scale=sa, b_scale=sw)
    rows = torch.cat([torch.tensor([0, 1, 127, 128, 255, 256, 257, 75_263, 75_264, 75_519, 75_520, 75_598, 75_599]),
                      torch.randint(0, M, (83,), generator=torch.Generator().manual_seed(K))]).to(DEV)
    A = (qa[rows].view(F8).double() * sa[rows, None].double()).cpu()
    W = (qw.view(F8).double() * sw[:, None].double()).cpu()
    ref = A @ W.t() + bias.double().cpu()
    mag = A.abs() @ W.abs().t()
    got = c[rows].double().cpu()
    err = (got - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -16 * mag + 1e-30
    assert bool(torch.isfinite(got).all())
    assert bool((err <= bound).all()), (K, what, float((err / bound).max()))
    # and almost every element IS the correctly rounded bf16 of the exact result
    assert (got != ref.to(BF).double()).double().mean().item() < 0.02
    # the whole output is live (no tile skipped): every 256-row x 256-column tile has a non-trivial checksum
    tiles = c[: (M // 256) * 256].view(M // 256, 256, N // 256, 256).float().abs().sum(dim=(1, 3))
    assert bool((tiles > 0).all()) and bool(torch.isfinite(tiles).all())
