"""The B-packed form of the default GEMM schedule (round 5): a linear layer's weight stored in MFMA-fragment order
(alg_pack_b_bf16) is loaded from L2 straight into the registers the MFMA reads -- no LDS-DMA, no ring slot, no fragment read for B.
Same MFMAs in the same order on the same values: every result must be BIT-IDENTICAL to the row-major call."""
import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def test_pack_b_layout():
    N, K = 100, 192                                   # N not a multiple of 32: the last n-block is zero-padded
    w = _rand((N, K), 1)
    pk = _lib.PackedB(w)
    assert pk.packed.shape == (128, K)
    flat = pk.packed.view(-1)
    wp = torch.zeros(128, K, dtype=BF, device=DEV)
    wp[:N] = w
    for nb, ks, lane in [(0, 0, 0), (0, 3, 37), (1, 11, 63), (3, 5, 31), (3, 11, 32), (2, 7, 5)]:
        row, k0 = nb * 32 + (lane & 31), ks * 16 + 8 * (lane >> 5)
        at = (nb * (K // 16) + ks) * 512 + lane * 8
        assert torch.equal(flat[at:at + 8], wp[row, k0:k0 + 8]), (nb, ks, lane)
    # as a whole: [nb][ks][h2][l31][8] <- [nb][l31][ks][h2][8]
    ref = wp.view(4, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)
    assert torch.equal(flat, ref)


@pytest.mark.parametrize("M,N,K,act,bias", [(300, 256, 256, 0, True), (1000, 3072, 3072, 0, True), (257, 104, 320, 0, False),
                                            (777, 1024, 832, _lib.ACT_GELU_TANH, True), (512, 12288, 3072, _lib.ACT_GELU_TANH, True),
                                            (640, 512, 256, _lib.ACT_SILU, True), (35552, 3072, 3072, 0, True)])
def test_packed_b_is_bit_identical_to_row_major(M, N, K, act, bias):
    a, w = _rand((M, K), 2), _rand((N, K), 3, 0.05)
    b = _rand((N,), 4) if bias else None
    pk = _lib.PackedB(w)
    assert pk.usable(K, None, None)
    c_rows = torch.full((M, N), 7.0, dtype=BF, device=DEV)
    c_pack = torch.full((M, N), -3.0, dtype=BF, device=DEV)
    _lib.gemm(a, w, c_rows, M, N, K, K, K, N, bias=b, act=act)
    _lib.gemm(a, pk, c_pack, M, N, K, K, K, N, bias=b, act=act)
    assert torch.equal(c_rows, c_pack)
    ref = a[:256].float() @ w.float().T + (b.float() if bias else 0.0)
    if act == 0:
        assert (c_pack[:256].float() - ref).abs().max().item() < 0.05 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kt", list(range(4, 16)))
def test_packed_b_every_loop_length(kt):
    """K / 64 = 4 .. 15: no steady-state pair, one, several, even and odd -- every path through the statement's prologue, loop and tails"""
    M, N, K = 520, 384, 64 * kt
    a, w = _rand((2, M, K), 5), _rand((N, K), 6, 0.1)
    pk = _lib.PackedB(w)
    c_rows = torch.zeros(2, M, N, dtype=BF, device=DEV)
    c_pack = torch.ones(2, M, N, dtype=BF, device=DEV)
    kw = dict(batch=2, strideA=M * K, strideC=M * N)
    _lib.gemm(a, w, c_rows, M, N, K, K, K, N, **kw)
    _lib.gemm(a, pk, c_pack, M, N, K, K, K, N, **kw)
    assert torch.equal(c_rows, c_pack)


def test_packed_b_in_the_pair_launch_and_fallbacks(monkeypatch):
    M, D = 700, 512
    y, wqk, wv = _rand((M, D), 7), _rand((2 * D, D), 8, 0.05), _rand((D, D), 9, 0.05)
    S_pad = 704
    def run(first_w):
        qk = torch.zeros(M, 2 * D, dtype=BF, device=DEV)
        vt = torch.zeros(D, S_pad, dtype=BF, device=DEV)
        _lib.gemm_pair(((y, first_w, qk, M, 2 * D, D, D, D, 2 * D), {}),
                       ((wv, y, vt, D, M, D, D, D, S_pad), dict(flags=_lib.GEMM_PERMUTE_COLS)))
        return qk, vt
    pk = _lib.PackedB(wqk)
    qk0, vt0 = run(wqk)
    qk1, vt1 = run(pk)
    assert torch.equal(qk0, qk1) and torch.equal(vt0, vt1)
    # a residual form cannot take the packed weight: the wrapper hands it the rows (same result as the row-major call)
    r = _rand((M, 2 * D), 10)
    c0, c1 = r.clone(), r.clone()
    _lib.gemm(y, wqk, c0, M, 2 * D, D, D, D, 2 * D, R=c0, ldr=2 * D)
    _lib.gemm(y, pk, c1, M, 2 * D, D, D, D, 2 * D, R=c1, ldr=2 * D)
    assert torch.equal(c0, c1)
    # the reference schedule reads rows
    monkeypatch.setenv("ALG_GEMM_PIPE", "6")
    _lib.reload_env()
    try:
        assert _lib.gemm_pipe() == 6 and not pk.usable(D, None, None)
        c2 = torch.zeros(M, 2 * D, dtype=BF, device=DEV)
        _lib.gemm(y, pk, c2, M, 2 * D, D, D, D, 2 * D)
        assert torch.equal(c2, qk0)
        args, _ = _lib.gemm_args(y, pk.packed, c2, M, 2 * D, D, D, D, 2 * D, flags=_lib.GEMM_B_PACKED)
        import ctypes
        assert _lib.load_library().alg_gemm_bf16(ctypes.byref(args), None) != 0       # the flag itself is refused on schedule 6
    finally:
        monkeypatch.delenv("ALG_GEMM_PIPE")
        _lib.reload_env()
