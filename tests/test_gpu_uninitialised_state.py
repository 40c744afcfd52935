"""The hand-written asm kernels of the product must not read state they never wrote.

A stress loop that re-launches ONE kernel never varies what the previous kernel left in the register file, in LDS or in the
scalar registers -- inside a forward every launch follows a DIFFERENT kernel.  tests/helpers/reg_poison.hip rewrites all 512
vector registers of every SIMD, the 160 KiB of LDS of every CU and s[16:99] + vcc with one bit pattern (NaN, huge, negative,
all-ones, zero) right in front of each launch; the result must equal the unpoisoned launch bit for bit.  (Round 4: built to test
the "uninitialised register" explanation of the shelved 64-queries-per-wave attention's first-round mismatches; that kernel passes
too -- round 4's poison probe, profiles/r4_poison_probe.jsonl -- so the explanation is excluded.)"""
import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def poison():
    from helpers import poison as P
    lib = P.load()
    # the poison must be visible to the next kernel, or the test below proves nothing
    out = torch.zeros(256 * 256 * 2, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.reg_poison(0x12345678, P.ALL, 512, st) == 0
    assert lib.reg_peek(out.data_ptr(), 256, st) == 0
    torch.cuda.synchronize()
    seen = (out.view(-1, 2) == 0x12345678).float().mean(dim=0)
    assert seen[0].item() > 0.99 and seen[1].item() > 0.99, seen     # v200 and a200 of a fresh wave hold the pattern
    return lib, P


def _cases():
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
    S = 4160
    cases = {}
    for name, Hh, d in (("flash_attn_d64_pipe", 48, 64), ("flash_attn_d128_q64 (default at this length)", 40, 128)):
        D, S_pad = Hh * d, (S + 63) // 64 * 64
        qk = rn(1, S, 2 * D)
        vt = torch.zeros(1, D, S_pad, dtype=BF, device="cuda")
        vt[:, :, :S] = rn(1, D, S)
        out = torch.empty(1, S, D, dtype=BF, device="cuda")
        if d == 64:
            qk.view(1, S, 2, D)[:, :, 0] *= 0.125 * 1.4426950408889634
            run = lambda qk=qk, vt=vt, out=out, Hh=Hh, D=D, S_pad=S_pad: _lib.flash_attn_d64(
                qk, qk, vt, out, 1, Hh, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D, q_prescaled=True)
        else:
            run = lambda qk=qk, vt=vt, out=out, Hh=Hh, D=D, S_pad=S_pad: _lib.flash_attn_d128(
                qk, qk, vt, out, 1, Hh, S, S, S * 2 * D, 2 * D, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 128 ** -0.5, k_off=D)
        cases[name] = (run, out)
    M, N, K = 2100, 1024, 1536
    a, w, bias, gate = rn(M, K), rn(N, K, sc=0.05), rn(N), rn(1, 2 * N, sc=0.5)
    x0 = rn(M, N)
    o1, o2 = torch.empty(M, N, dtype=BF, device="cuda"), torch.empty(M, N, dtype=BF, device="cuda")
    cases["gemm schedule 9 (GELU)"] = (lambda: _lib.gemm(a, w, o1, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH), o1)

    def res():
        o2.copy_(x0)
        _lib.gemm(a, w, o2, M, N, K, K, K, N, bias=bias, R=o2, ldr=N, gate=gate, strideGate=2 * N, seg_split=226)
    cases["gemm schedule 9 (gate * x + residual, in-loop residual fetch)"] = (res, o2)
    # e4m3 operands on schedule 9 (round 4), plain and residual + fp32 gate forms
    qa = torch.randint(0, 255, (M, K), dtype=torch.uint8, device="cuda")
    qw = torch.randint(0, 255, (N, K), dtype=torch.uint8, device="cuda")
    for t in (qa, qw):
        t[t == 0x7F] = 0x3F
        t[t == 0xFF] = 0xBF
    sa, sw = torch.rand(M, device="cuda") * 1e-2, torch.rand(N, device="cuda") * 1e-2
    o3, o4 = torch.empty(M, N, dtype=BF, device="cuda"), torch.empty(M, N, dtype=BF, device="cuda")
    g32 = torch.randn(1, 2 * N, generator=g, device="cuda")
    cases["fp8 gemm schedule 9 (plain)"] = (lambda: _lib.gemm(qa, qw, o3, M, N, K, K, K, N, bias=bias, a_scale=sa, b_scale=sw), o3)

    def res8():
        o4.copy_(x0)
        _lib.gemm(qa, qw, o4, M, N, K, K, K, N, bias=bias, R=o4, ldr=N, gate=g32, strideGate=2 * N, seg_split=1 << 30,
                  flags=_lib.GEMM_GATE_F32, a_scale=sa, b_scale=sw)
    cases["fp8 gemm schedule 9 (fp32 gate * x + residual)"] = (res8, o4)
    yq, wqk, wv, bv = rn(1, M, K), rn(2 * K, K, sc=0.05), rn(K, K, sc=0.05), rn(K)
    Mp = (M + 63) // 64 * 64
    qkb, vtb = torch.empty(1, M, 2 * K, dtype=BF, device="cuda"), torch.zeros(1, K, Mp, dtype=BF, device="cuda")
    wq, bq = 1 + rn(64, sc=0.2), rn(64, sc=0.2)
    cos, sin = torch.rand(M - 17, 64, device="cuda"), torch.rand(M - 17, 64, device="cuda")
    cases["gemm pair + QK norm / rope store loop"] = (lambda: _lib.gemm_pair_qk(
        ((yq, wqk, qkb, M, 2 * K, K, K, K, 2 * K), dict(batch=1, strideA=M * K, strideC=M * 2 * K)),
        ((wv, yq, vtb, K, M, K, K, K, Mp), dict(bias=bv, batch=1, strideB=M * K, strideC=K * Mp,
                                               flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)),
        wq, bq, wq, bq, cos, sin, K // 64, 17, 1e-6, q_scale=0.18), qkb)
    return cases


def test_asm_kernels_ignore_what_the_previous_kernel_left_behind(poison, monkeypatch):
    lib, P = poison
    st = torch.cuda.current_stream().cuda_stream
    cases = _cases()
    cases["flash_attn_d128_pipe"] = cases["flash_attn_d128_q64 (default at this length)"] + ("ALG_ATTN128_Q64", "0")
    for name, case in cases.items():
        run, out = case[:2]
        if len(case) > 2:
            monkeypatch.setenv(case[2], case[3])
        run()
        torch.cuda.synchronize()
        ref = out.clone()
        run()
        assert torch.equal(out, ref), name + ": not deterministic without poison"
        for parts, pat in [(P.ALL, "nan"), (P.ALL, "big"), (P.ALL, "neg"), (P.ALL, "allbits"), (P.ALL, "zero"),
                           (P.VGPR_LO | P.VGPR_HI, "nan"), (P.AGPR_LO | P.AGPR_HI, "nan"), (P.LDS, "nan"), (P.SGPR, "alt")]:
            for _ in range(4):
                assert lib.reg_poison(P.PATTERNS[pat], parts, 512, st) == 0
                run()
                assert torch.equal(out, ref), "%s: differs behind poison parts=%d pattern=%s" % (name, parts, pat)
