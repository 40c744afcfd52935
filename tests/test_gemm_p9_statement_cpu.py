"""GEMM schedule 9's generated main-loop statements (scripts/gen_gemm_p9.py -> alg_amd/csrc/gemm_p9_loop.inc: the plain form and the
residual form) checked AS PROGRAMS on the CPU, like the attention statements (test_attn_q64_statement_cpu.py): the instruction-level
emulator runs the asm text for the four waves of one 256 x 256 tile the way gemm_kernel.h's frame drives it -- ten-slot LDS ring,
LDS-DMA two k-tiles ahead, fragment reads one k-step ahead, one counted wait and one barrier per k-tile, the residual tile fetched
inside the loop with its catch-up chain for short K -- under the weakest memory ordering the ISA allows (fragment reads and DMA /
buffer loads land only at the counted wait that covers them), and the accumulators are compared with a float64 A B^T; the harness is
shown to catch seeded defects.  The e4m3 loop (gen_gemm_p9_fp8.py, its block-scaled MFMA emulated with all scales 2^0) runs in the same
harness.  No GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gemm_emu as H  # noqa: E402

MODES = [(True, False), (False, True), (True, True)]     # (lazy fragment reads, lazy DMA / buffer loads)
TOL = 2e-6      # fp32 accumulation of exact bf16 products over K <= 960


def relerr(c, ref):
    return float(np.abs(c - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("nk", [2, 3, 4, 7, 12])
def test_plain_statement_computes_the_tile_under_the_weakest_memory_ordering(nk):
    """K / 64 = 2 (no steady-state trip), 3, 4, 7 and 12 (the ten-slot ring wraps twice)"""
    pb = H.Problem(nk, seed=nk)
    ref = pb.reference()
    for lazy_reads, lazy_dma in MODES:
        c, _, _ = H.run_plain(pb, lazy_reads, lazy_dma)
        assert relerr(c, ref) < TOL, (nk, lazy_reads, lazy_dma, relerr(c, ref))


def test_plain_statement_on_a_ragged_tile_never_reads_outside_the_panels():
    """200 valid A rows, 130 valid B rows: the DMA's row index is clamped to the last valid row (rmaxa / rmaxb); everything outside
    the two panels is NaN in the memory image and a fetch outside the image is a fault"""
    pb = H.Problem(6, seed=21, rows_a=200, rows_b=130)
    c, _, _ = H.run_plain(pb, True, True)
    assert np.isfinite(c).all() and relerr(c, pb.reference()) < TOL


@pytest.mark.parametrize("nk", [2, 3, 5, 9, 10, 11, 13])
def test_residual_statement_returns_the_tile_and_the_residual(nk):
    """K / 64 - 2 = 0 .. 11 steady-state trips: every entry of the catch-up chain that fetches what the loop did not get to (the loop
    fetches four residual quads per k-tile over its first eight trips), rows past M read as zeros through the descriptor"""
    rows = 256 if nk % 2 else 216
    pb = H.Problem(nk, seed=100 + nk, rows_a=rows)
    ref = pb.reference()
    rref = np.zeros((256, 256))
    rref[:rows] = pb.r
    for lazy_reads, lazy_dma in (MODES if nk in (2, 10) else MODES[2:]):
        c, r, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, res=True)
        assert relerr(c, ref) < TOL, (nk, relerr(c, ref))
        assert np.array_equal(r, rref), nk


def _replace_all(lines, old, new):
    assert any(ln == old for ln in lines), old
    return [(new if ln == old else ln) for ln in lines]


MUTATIONS = {
    "DMA wait four too loose": lambda L: _replace_all(L, "s_waitcnt vmcnt(8) lgkmcnt(0)", "s_waitcnt vmcnt(12) lgkmcnt(0)"),
    "no barrier": lambda L: [ln for ln in L if ln != "s_barrier"],
    "fragment waits dropped": lambda L: [ln for ln in L if not (ln.startswith("s_waitcnt lgkmcnt(") and ln != "s_waitcnt lgkmcnt(0)")],
    "B DMA into the wrong slot": lambda L: [ln.replace("s_add_u32 m0, %[t8], 16384", "s_add_u32 m0, %[t8], 32768") for ln in L],
    "prologue does not wait for the first k-tile": lambda L: _replace_all(L, "s_waitcnt vmcnt(16)", "s_waitcnt vmcnt(32)"),
}


@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_the_harness_sees_seeded_defects_in_the_gemm_loop(name):
    pb = H.Problem(7, seed=7)
    ref = pb.reference()
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        try:
            c, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, mutate=MUTATIONS[name])
            e = relerr(c, ref)
            worst = max(worst, e if np.isfinite(e) else 1.0)
        except RuntimeError:          # deadlock / runaway / memory fault: also a detection
            worst = 1.0
    assert worst > 1e3 * TOL, (name, worst)


def test_residual_wait_is_load_bearing():
    """the residual form lets twelve instead of eight loads stay in flight at the barrier of a k-tile that also fetched residual quads;
    one more and a DMA of the NEXT k-tile is not covered"""
    pb = H.Problem(6, seed=9)
    ref = pb.reference()
    mut = lambda L: _replace_all(L, "s_waitcnt vmcnt(12) lgkmcnt(0)", "s_waitcnt vmcnt(16) lgkmcnt(0)")
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        c, _, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, mutate=mut, res=True)
        e = relerr(c, ref)
        worst = max(worst, e if np.isfinite(e) else 1.0)
    assert worst > 1e3 * TOL


# ---------------------------------------------------------------------------------------------------------------------
# the e4m3 loop (scripts/gen_gemm_p9_fp8.py -> gemm_p9_fp8_loop.inc): same ring and DMA protocol, ONE set of 8-register operands that
# is re-read for the next k-pair right behind its last MFMA, v_mfma_scale_f32_32x32x64_f8f6f4 with all block scales 2^0
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nk,res", [(2, False), (3, False), (7, False), (12, False), (2, True), (6, True), (13, True)])
def test_fp8_statements_compute_the_tile_under_the_weakest_memory_ordering(nk, res):
    rows = 256 if nk % 2 == 0 else 200
    pb = H.ProblemFp8(nk, seed=50 + nk, rows_a=rows)
    ref = pb.reference()
    for lazy_reads, lazy_dma in (MODES if nk in (2, 7) else MODES[2:]):
        out = H.run_plain(pb, lazy_reads, lazy_dma, res=res)
        assert relerr(out[0], ref) < TOL, (nk, res, lazy_reads, lazy_dma, relerr(out[0], ref))
        if res:
            rref = np.zeros((256, 256))
            rref[:rows] = pb.r
            assert np.array_equal(out[1], rref)


def _reread_one_mfma_early(lines):
    """the first operand re-read that sits right behind an MFMA moves in front of it: the MFMA then multiplies the NEXT k-pair's bytes"""
    out = list(lines)
    start = out.index("1:")                                   # inside the steady-state loop
    for i in range(start, len(out) - 1):
        if out[i].startswith("v_mfma") and out[i + 1].startswith("ds_read_b128"):
            dst = out[i + 1].split()[1].rstrip(",")           # v[a:b]
            lo = int(dst[2:].split(":")[0])
            ops = [t.strip() for t in out[i].split(None, 1)[1].split(",")]
            for src in ops[1:3]:
                a, b = (int(x) for x in src[2:-1].split(":"))
                if a <= lo <= b:
                    out[i], out[i + 1] = out[i + 1], out[i]
                    return out
    raise AssertionError("no re-read found behind the MFMA that reads its operand")


FP8_MUTATIONS = {
    "operand re-read one MFMA early": _reread_one_mfma_early,
    "no barrier": lambda L: [ln for ln in L if ln != "s_barrier"],
    "DMA wait four too loose": lambda L: [ln.replace("s_waitcnt vmcnt(8)", "s_waitcnt vmcnt(12)") for ln in L],
}


@pytest.mark.parametrize("name", sorted(FP8_MUTATIONS))
def test_the_harness_sees_seeded_defects_in_the_fp8_loop(name):
    pb = H.ProblemFp8(7, seed=8)
    ref = pb.reference()
    worst = 0.0
    for lazy_reads, lazy_dma in MODES + [(False, False)]:
        try:
            c = H.run_plain(pb, lazy_reads, lazy_dma, mutate=FP8_MUTATIONS[name])[0]
            e = relerr(c, ref)
            worst = max(worst, e if np.isfinite(e) else 1.0)
        except RuntimeError:
            worst = 1.0
    assert worst > 1e3 * TOL, (name, worst)
