"""GEMM schedule 10 (round 6): schedule 9's tile, LDS ring, DMA and barrier protocol on v_mfma_f32_16x16x32_bf16 -- the bf16 MFMA
shape that sustains ~10 % more than 32x32x16 under the package power cap (scripts/micro/mfma_shape.hip).  The hardware sums 32
instead of 16 products per instruction, so its fp32 rounding points differ from schedules 9 / 6: the results are NOT bit-identical
to theirs, and this file is the net in place of the bit-identity tests those two share (test_gpu_dit_kernels.py):
  * every epilogue form (plain + column bias, GELU, the transposed / permuted V^T projection with its per-row bias, residual with no
    gate / bf16 gate across a segment boundary / fp32 gate / the straddled generic loop) at K / 64 = 2 .. 13 (every entry of the
    residual catch-up chain), with edge tiles in M and N, against a float64 matmul -- never further from it than schedule 9 is --
    and against schedule 9 element by element (one bf16 rounding step of the linear's output);
  * N % 8 != 0 (the element-exact epilogue, which has its own (register, lane) -> (row, column) map for the 16 x 16 blocks);
  * the pair launch and the fused QK LayerNorm + rope store loop: bit-identical to the separate launches ON THIS SCHEDULE;
  * the C2 shapes, where a wrong fragment address would show as a wrong matrix.
The statement itself runs as a program on the CPU (tests/test_gemm_p10_statement_cpu.py).  Reference call sites: the nn.Linear
layers behind /root/reference/pipeline_cogvideox_image2video_lowpass.py:1082-1090."""
import pytest
import torch
import torch.nn.functional as F

from alg_amd import _lib

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
FORMS = ["plain", "gelu", "vt", "res", "res_gate_seg", "res_gate_f32", "res_gate_f32_straddle"]


def swap23(n):
    return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1)


def _run(form, a, w, bias, x0, gate, gate32, brow, M, N, K):
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


def _reference(form, a, w, bias, x0, gate, gate32, brow, M, N, K):
    """float64 restatement of the epilogue forms with the product's rounding points (bf16 linear output, bf16 gate product)"""
    lin = a.double() @ w.double().t()
    if form == "vt":
        npad = (N + 63) // 64 * 64
        out = torch.zeros(M, npad, dtype=torch.float64, device="cuda")
        perm = torch.tensor([swap23(n) for n in range(N)], device="cuda")
        out[:, perm] = lin + brow.double()[:, None]
        return out
    lin = lin + bias.double()
    if form == "plain":
        return lin
    if form == "gelu":
        return F.gelu(lin.to(BF).double(), approximate="tanh")
    x = lin.to(BF).double()
    rows = torch.arange(M, device="cuda")[:, None]
    if form == "res":
        return x0.double() + x
    if form == "res_gate_seg":
        g = torch.where(rows < M // 3, gate[:, :N].double(), gate[:, N:].double())
        return x0.double() + (g * x).to(BF).double()
    if form == "res_gate_f32":
        return x0.double() + gate32[:, :N].double() * x
    g = torch.where(rows < 100, gate32[:, :N].double(), gate32[:, N:].double())
    return x0.double() + g * x


@pytest.mark.parametrize("form", FORMS)
def test_schedule10_meets_float64_like_schedule9_in_every_epilogue_form(monkeypatch, form):
    g = torch.Generator(device="cuda").manual_seed(10)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    for M, N, K in [(300, 520, 64 * k) for k in range(2, 14)] + [(1111, 96, 3072), (2100, 1024, 64 * 23)]:
        a, w, bias, x0 = rn(M, K), rn(N, K, sc=0.05), rn(N), rn(M, N)
        gate, gate32, brow = rn(1, 2 * N, sc=0.5), torch.randn(1, 2 * N, generator=g, device="cuda"), rn(M)
        args = (form, a, w, bias, x0, gate, gate32, brow, M, N, K)
        ref = _reference(*args)
        monkeypatch.setenv("ALG_GEMM_PIPE", "9")
        got9 = _run(*args).double()
        monkeypatch.setenv("ALG_GEMM_PIPE", "10")
        got10 = _run(*args).double()
        assert torch.equal(_run(*args).double(), got10), (form, M, N, K)            # deterministic
        scale = ref.abs().max().item()
        e9, e10 = (got9 - ref).abs(), (got10 - ref).abs()
        # never further from the exact result than schedule 9 (both are one bf16 rounding of an fp32 sum of K products)
        assert e10.max().item() <= 1.25 * e9.max().item() + 2e-3 * scale, (form, M, N, K, e10.max().item(), e9.max().item())
        assert e10.pow(2).mean().sqrt().item() <= 1.05 * e9.pow(2).mean().sqrt().item() + 1e-5 * scale, (form, M, N, K)
        # element by element against schedule 9: the linear's bf16 output may round the other way (one step, 2^-8 relative), and
        # the residual forms add that step times the gate
        d = (got10 - got9).abs()
        lin_max = (a.double() @ w.double().t()).abs().max()
        step = 2.0 ** -7 * torch.maximum(got9.abs(), got10.abs()) + 2.0 ** -7 * 5.0 * lin_max      # (5: the largest |gate|)
        assert bool((d <= step).all()), (form, M, N, K, d.max().item())
        assert (d > 0).double().mean().item() < 0.25, (form, M, N, K)               # most elements round the same way


@pytest.mark.parametrize("M,N,K", [(300, 250, 192), (33, 6, 128), (513, 1001, 256), (70, 52, 3072)])
def test_schedule10_element_exact_epilogue(monkeypatch, M, N, K):
    """N % 8 != 0 (and N % 4 != 0): the tile leaves through the element-exact epilogue; bias, GELU, residual with a bf16 gate"""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    a, w, bias, x0, gate = rn(M, K), rn(N, K, sc=0.05), rn(N), rn(M, N), rn(1, 2 * N, sc=0.5)
    outs = {}
    for pipe in ("9", "10"):
        monkeypatch.setenv("ALG_GEMM_PIPE", pipe)
        c = torch.full((M, N), 7.0, dtype=BF, device="cuda")
        _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH)
        x = x0.clone()
        _lib.gemm(a, w, x, M, N, K, K, K, N, bias=bias, R=x, ldr=N, gate=gate, strideGate=2 * N, seg_split=M // 2)
        outs[pipe] = (c.double(), x.double())
    lin = (a.double() @ w.double().t() + bias.double())
    ref_c = F.gelu(lin.to(BF).double(), approximate="tanh")
    rows = torch.arange(M, device="cuda")[:, None]
    gsel = torch.where(rows < M // 2, gate[:, :N].double(), gate[:, N:].double())
    ref_x = x0.double() + (gsel * lin.to(BF).double()).to(BF).double()
    for k, ref in enumerate((ref_c, ref_x)):
        e9, e10 = (outs["9"][k] - ref).abs().max().item(), (outs["10"][k] - ref).abs().max().item()
        assert e10 <= 1.25 * e9 + 2e-3 * ref.abs().max().item(), (k, e10, e9)


def test_schedule10_is_transpose_detecting_and_exact_on_single_term_sums(monkeypatch):
    """A = I with an asymmetric B: every output is ONE product -- exact on any schedule, wrong under a swapped fragment or block map"""
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    for M, N, K in [(256, 256, 256), (512, 768, 512)]:
        A = torch.zeros(M, K)
        A[torch.arange(M), (torch.arange(M) * 7) % K] = 1.0
        B = (torch.arange(N)[:, None] * 0.25 + torch.arange(K)[None, :] * 0.001953125).to(BF)
        C = torch.empty(M, N, dtype=BF, device="cuda")
        _lib.gemm(A.to(BF).cuda(), B.cuda(), C, M, N, K, K, K, N)
        assert torch.equal(C.cpu().float(), B.float()[:, (torch.arange(M) * 7) % K].t())


@pytest.mark.parametrize("S,D,N", [(17776, 3072, 2), (300, 512, 3), (257, 128, 2)])
def test_schedule10_pair_launches_are_bit_identical_to_their_parts(monkeypatch, S, D, N):
    """under ALG_GEMM_PIPE=10: alg_gemm_bf16_pair == the two launches; alg_gemm_bf16_pair_qk == pair + alg_qk_norm_rope_scaled"""
    monkeypatch.setenv("ALG_GEMM_PIPE", "10")
    heads = D // 64
    g = torch.Generator(device="cuda").manual_seed(S + D)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    S_pad = (S + 63) // 64 * 64
    T = min(226, S // 3)
    y, wqk, bqk, wv, bv = rn(N, S, D), rn(2 * D, D, sc=0.05), rn(2 * D), rn(D, D, sc=0.05), rn(D)
    wq, bq, wk, bk = (1 + rn(64, sc=0.2)), rn(64, sc=0.2), (1 + rn(64, sc=0.2)), rn(64, sc=0.2)
    ang = torch.rand(S - T, 32, generator=g, device="cuda") * 6.28
    cos, sin = ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous()

    def calls(qk, vt):
        return (((y, wqk, qk, S, 2 * D, D, D, D, 2 * D), dict(bias=bqk, batch=N, strideA=S * D, strideC=S * 2 * D)),
                ((wv, y, vt, D, S, D, D, D, S_pad), dict(bias=bv, batch=N, strideB=S * D, strideC=D * S_pad,
                                                        flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)))

    def fresh():
        return torch.full((N, S, 2 * D), 7.0, dtype=BF, device="cuda"), torch.zeros(N, D, S_pad, dtype=BF, device="cuda")

    qk0, vt0 = fresh()
    for a_, kw in calls(qk0, vt0):
        _lib.gemm(*a_, **kw)
    qk1, vt1 = fresh()
    _lib.gemm_pair(*calls(qk1, vt1))
    assert torch.equal(qk1, qk0) and torch.equal(vt1, vt0)
    _lib.qk_norm_rope_(qk0, wq, bq, wk, bk, cos, sin, N, S, heads, T, 1e-6, q_scale=0.18033688)
    if heads % 4 == 0:
        qk2, vt2 = fresh()
        _lib.gemm_pair_qk(*calls(qk2, vt2), wq, bq, wk, bk, cos, sin, heads, T, 1e-6, q_scale=0.18033688)
        assert torch.equal(qk2, qk0) and torch.equal(vt2, vt0)
    # against float64 at this shape: the Q|K projection (sampled rows)
    rows = torch.tensor([0, 1, 15, 16, 17, 255, 256, S // 2, S - 1], device="cuda")
    ref = y[N - 1, rows].double() @ wqk.double().t() + bqk.double()
    assert ((qk1[N - 1, rows].double() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()).item()
