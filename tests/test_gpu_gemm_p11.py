"""GEMM schedule 11 (round 6): 1 x 4 wave layout on v_mfma_f32_16x16x32_bf16, the weight operand pre-packed in
MFMA-fragment order (alg_pack_b_p11 / _lib.PackedB) and loaded straight from L2 into registers, A through the LDS ring.  Per output
element it sums the same products in the same order as schedule 10 (k-steps of 32 in ascending order, the same fragment k-map), so the
net is BIT-IDENTITY with schedule 10's row-major call: every epilogue form (plain + column bias, GELU, SiLU, residual with no gate / bf16
gate across a segment boundary / fp32 gate / the straddled generic loop), K / 64 = 2 .. 13 (both parities of the two-set B rotation,
every entry of the residual catch-up chain), edge tiles in M and N (packed rows past N are zeros, stores are guarded), batched A with a
shared weight, N % 8 != 0 (element-exact epilogue), the C2 shapes; bad calls are rejected before a launch.  The statement itself runs
as a program on the CPU (tests/test_gemm_p11_statement_cpu.py).  Reference call sites: the nn.Linear layers behind
/root/reference/pipeline_cogvideox_image2video_lowpass.py:1082-1090."""
import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
FORMS = ["plain", "gelu", "silu", "res", "res_gate_seg", "res_gate_f32", "res_gate_f32_straddle"]


def _run(form, a, w, bias, x0, gate, gate32, M, N, K, batch=1):
    kw = dict(batch=batch, strideA=M * K, strideC=M * N) if batch > 1 else {}
    if form in ("plain", "gelu", "silu"):
        c = torch.full((batch, M, N), 7.0, dtype=BF, device="cuda")
        act = {"plain": _lib.ACT_NONE, "gelu": _lib.ACT_GELU_TANH, "silu": _lib.ACT_SILU}[form]
        _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias, act=act, **kw)
        return c
    x = x0.clone()
    if batch > 1:
        kw.update(strideR=M * N)
    if form == "res_gate_seg":
        kw.update(gate=gate, strideGate=2 * N if batch > 1 else 0, seg_split=M // 3)
    elif form == "res_gate_f32":
        kw.update(gate=gate32, strideGate=2 * N if batch > 1 else 0, seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)
    elif form == "res_gate_f32_straddle":
        kw.update(gate=gate32, strideGate=2 * N if batch > 1 else 0, seg_split=100, flags=_lib.GEMM_GATE_F32)
    _lib.gemm(a, w, x, M, N, K, K, K, N, bias=bias, R=x, ldr=N, **kw)
    return x


@pytest.mark.parametrize("form", FORMS)
def test_schedule11_is_bit_identical_to_schedule10_in_every_epilogue_form(monkeypatch, form):
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    for M, N, K in [(300, 520, 64 * k) for k in range(2, 14)] + [(1111, 96, 3072), (2100, 1024, 64 * 23), (257, 256, 128)]:
        a, w, bias, x0 = rn(1, M, K), rn(N, K, sc=0.05), rn(N), rn(1, M, N)
        gate, gate32 = rn(1, 2 * N, sc=0.5), torch.randn(1, 2 * N, generator=g, device="cuda")
        want = _run(form, a, w, bias, x0, gate, gate32, M, N, K)
        pk = _lib.PackedB(w)
        for _ in range(2):
            got = _run(form, a, pk, bias, x0, gate, gate32, M, N, K)
            if not torch.equal(got, want):
                d = (got.float() - want.float()).abs()
                raise AssertionError("%s M=%d N=%d K=%d: %d elements differ, max %.4g, first %s" % (
                    form, M, N, K, int((d > 0).sum()), d.max().item(), (d > 0).nonzero()[:3].tolist()))


@pytest.mark.parametrize("form", ["plain", "res_gate_seg", "res_gate_f32"])
def test_schedule11_batched_activations_shared_weight(monkeypatch, form):
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    g = torch.Generator(device="cuda").manual_seed(12)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    nb, M, N, K = 3, 290, 768, 512
    a, w, bias, x0 = rn(nb, M, K), rn(N, K, sc=0.05), rn(N), rn(nb, M, N)
    gate, gate32 = rn(nb, 2 * N, sc=0.5), torch.randn(nb, 2 * N, generator=g, device="cuda")
    want = _run(form, a, w, bias, x0, gate, gate32, M, N, K, batch=nb)
    got = _run(form, a, _lib.PackedB(w), bias, x0, gate, gate32, M, N, K, batch=nb)
    assert torch.equal(got, want)


def test_schedule11_strided_operands_and_offsets(monkeypatch):
    """the addressing forms the HunyuanVideo blocks use (not on schedule 11 in the product, but a PackedB must be safe wherever a weight is):
    A as a column window of a wider buffer (lda > K, a_off), C into a column window (ldc > N, c_off), the residual at a row offset, the
    per-segment gate of token replacement (gate_seg_stride), batch strides that are not M * ld"""
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    g = torch.Generator(device="cuda").manual_seed(14)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    nb, J, S, D, Mff = 2, 333, 300, 256, 512
    AM = D + Mff
    am, y = rn(nb, J, AM), rn(nb, J, D)
    w_o, b_o, w_f1, b_f1 = rn(D, D, sc=0.05), rn(D), rn(Mff, D, sc=0.05), rn(Mff)
    mod = rn(nb, 2, 6 * D, sc=0.5)
    x0 = rn(nb, J, D)

    def run(wo, wf1):
        x, h = x0.clone(), am.clone()
        # out-projection of the latent rows: A = the first D columns of the [J][D + Mff] buffer, gate per segment at a stride
        _lib.gemm(h, wo, x, S, D, D, AM, D, D, bias=b_o, R=x, ldr=D, gate=mod, gate_off=2 * D, strideGate=12 * D, gate_seg_stride=6 * D,
                  seg_split=100, batch=nb, strideA=J * AM, strideC=J * D, strideR=J * D)
        # ... of the prompt rows: row offsets on A, C and R
        _lib.gemm(h, wo, x, J - S, D, D, AM, D, D, bias=b_o, R=x, ldr=D, gate=mod, gate_off=2 * D, strideGate=12 * D, gate_seg_stride=0,
                  batch=nb, strideA=J * AM, strideC=J * D, strideR=J * D, a_off=S * AM, c_off=S * D, r_off=S * D)
        # ff1 with GELU into the column window [D, D + Mff) of the wide buffer
        _lib.gemm(y, wf1, h, S, Mff, D, D, D, AM, bias=b_f1, act=_lib.ACT_GELU_TANH, batch=nb, strideA=J * D, strideC=J * AM, c_off=D)
        return x, h

    want = run(w_o, w_f1)
    got = run(_lib.PackedB(w_o), _lib.PackedB(w_f1))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(got[1][:, :, :D], am[:, :, :D]) and torch.equal(got[1][:, S:], am[:, S:])    # nothing outside the windows is written


@pytest.mark.parametrize("M,N,K", [(300, 250, 192), (33, 6, 128), (513, 1001, 256)])
def test_schedule11_element_exact_epilogue(monkeypatch, M, N, K):
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    g = torch.Generator(device="cuda").manual_seed(M + N)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    a, w, bias, x0, gate = rn(1, M, K), rn(N, K, sc=0.05), rn(N), rn(1, M, N), rn(1, 2 * N, sc=0.5)
    pk = _lib.PackedB(w)
    for form in ("gelu", "res_gate_seg"):
        want = _run(form, a, w, bias, x0, gate, None, M, N, K)
        assert torch.equal(_run(form, a, pk, bias, x0, gate, None, M, N, K), want), form


def test_schedule11_at_the_c2_shapes_and_against_float64(monkeypatch):
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    g = torch.Generator(device="cuda").manual_seed(13)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    S, D = 17776, 3072
    y, h = rn(2, S, D), rn(2, S, 4 * D)
    for name, a, N, K in (("qk", y, 2 * D, D), ("ff1", y, 4 * D, D), ("ff2", h, D, 4 * D)):
        w, bias = rn(N, K, sc=0.02), rn(N, sc=0.02)
        c0 = torch.empty(2, S, N, dtype=BF, device="cuda")
        c1 = torch.empty_like(c0)
        kw = dict(bias=bias, batch=2, strideA=S * K, strideC=S * N)
        _lib.gemm(a, w, c0, S, N, K, K, K, N, **kw)
        _lib.gemm(a, _lib.PackedB(w), c1, S, N, K, K, K, N, **kw)
        assert torch.equal(c0, c1), name
        rows = torch.tensor([0, 1, 15, 16, 255, 256, S // 2, S - 1], device="cuda")
        ref = a[1, rows].double() @ w.double().t() + bias.double()
        assert ((c1[1, rows].double() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()).item(), name


def test_schedule11_rejects_what_it_cannot_take():
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(256, 256, generator=g, device="cuda") * 0.05).to(BF)
    a = torch.randn(64, 256, generator=g, device="cuda").to(BF)
    c = torch.empty(64, 256, dtype=BF, device="cuda")
    pk = _lib.PackedB(w)
    with pytest.raises(_lib.AlgHipError):                      # built for another K
        _lib.gemm(a, pk, c, 64, 256, 128, 256, 256, 256)
    with pytest.raises(_lib.AlgHipError):                      # a per-row bias belongs to the transposed V projection (B = activations)
        _lib.gemm(a, pk, c, 64, 256, 256, 256, 256, 256, bias=torch.zeros(64, dtype=BF, device="cuda"), flags=_lib.GEMM_BIAS_PER_ROW)
    with pytest.raises(_lib.AlgHipError):                      # K = 64: the asm loop needs two k-tiles
        _lib.PackedB(w[:, :64].contiguous()) and _lib.gemm(a, _lib.PackedB(w[:, :64].contiguous()), c, 64, 256, 64, 256, 64, 256)
