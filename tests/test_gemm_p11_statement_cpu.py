"""GEMM schedule 11's generated main-loop statements (scripts/gen_gemm_p11.py -> alg_amd/csrc/gemm_p11_loop.inc: 1 x 4 wave layout on
v_mfma_f32_16x16x32_bf16, A through the LDS ring, the weight operand pre-packed in fragment order and loaded straight into
registers -- VERDICT r5 item 1a) checked AS PROGRAMS on the CPU: four waves of one 256 x 256 tile in the instruction-level emulator
under lazy fragment reads / lazy vector-memory completion (LDS-DMA pieces AND the B loads land only at the counted wait that covers
them), the accumulators against a float64 A B^T, the residual tile bit for bit, both parities of the two-set B rotation (even and odd
steady-state counts, every residual exit), and seeded defects the harness has to catch.  No GPU."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gemm_emu as H  # noqa: E402
import gen_gemm_p11 as G11  # noqa: E402

MODES = [(True, False), (False, True), (True, True)]
TOL = 2e-6


def relerr(c, ref):
    return float(np.abs(c - ref).max() / np.abs(ref).max())


def reference(pb):
    """schedule 11's B rows past the operand's end are ZERO in the packed panel (the 2 x 2 schedules clamp the DMA row instead)"""
    ref = pb.reference()
    ref[:, pb.rows_b:] = 0.0
    return ref


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 8, 13])
def test_plain_statement_computes_the_tile_under_the_weakest_memory_ordering(nk):
    """K / 64 = 2 (no steady-state k-tile), 3 (one: the odd path), 4 (one pair), 5, 8, 13 (the ring wraps; both tail parities)"""
    pb = H.Problem(nk, seed=nk)
    ref = reference(pb)
    for lazy_reads, lazy_dma in (MODES if nk in (3, 8) else MODES[2:]):
        c, _, _ = H.run_p11(pb, lazy_reads, lazy_dma)
        assert relerr(c, ref) < TOL, (nk, lazy_reads, lazy_dma, relerr(c, ref))


def test_plain_statement_on_a_ragged_tile_never_reads_outside_the_panels():
    pb = H.Problem(6, seed=21, rows_a=200, rows_b=130)
    c, _, _ = H.run_p11(pb, True, True)
    assert np.isfinite(c).all() and relerr(c, reference(pb)) < TOL


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 6, 9, 10, 11, 12, 13])
def test_residual_statement_returns_the_tile_and_the_residual(nk):
    """K / 64 - 2 = 0 .. 11 steady-state k-tiles: every entry of the catch-up chain, leaving it with either parity"""
    rows = 256 if nk % 2 else 216
    pb = H.Problem(nk, seed=100 + nk, rows_a=rows)
    ref = reference(pb)
    rref = np.zeros((256, 256))
    rref[:rows] = pb.r
    for lazy_reads, lazy_dma in (MODES if nk in (2, 10) else MODES[2:]):
        c, r, _, _ = H.run_p11(pb, lazy_reads, lazy_dma, res=True)
        assert relerr(c, ref) < TOL, (nk, relerr(c, ref))
        assert np.array_equal(r, rref), nk


def test_the_text_in_the_tree_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "p11.inc"
    os.environ["P11_OUT"] = str(out)
    try:
        G11.main()
    finally:
        del os.environ["P11_OUT"]
    with open(os.path.join(ROOT, "alg_amd", "csrc", "gemm_p11_loop.inc")) as f:
        assert f.read() == out.read_text()


def test_statement_shape_per_k_tile():
    """per steady-state k-tile and wave: 128 MFMAs, 32 A fragment reads, 8 B loads, 8 LDS-DMA pieces (half of schedule 10's), one barrier"""
    L = G11.emit()
    body = L[L.index("1:"):L.index("s_branch 1b")]
    count = lambda pre: sum(1 for ln in body if ln.startswith(pre))
    assert count("v_mfma_f32_16x16x32_bf16") == 256 and count("ds_read_b128") == 64
    assert count("buffer_load_dwordx4") == 16 and count("global_load_lds_dwordx4") == 16 and count("s_barrier") == 2
    acc = [int(re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):", ln).group(1)) for ln in body if ln.startswith("v_mfma")]
    assert sorted(acc) == sorted(4 * list(range(0, 256, 4)))          # two k-tiles x two k-steps per accumulator block
    for a, b in zip(body, body[1:]):
        assert not (a.startswith("s_add_u32 m0") and b.startswith("global_load_lds"))
    # a k-tile's MFMAs read ONE B set (v[192:223] / v[224:255]), its B loads write the other
    n_mfma, halves = 0, [[], []]
    for ln in body:
        if ln.startswith("v_mfma"):
            halves[n_mfma // 128].append(("rd", int(re.search(r", v\[(\d+):\d+\], v\[", ln).group(1))))
            n_mfma += 1
        elif ln.startswith("buffer_load_dwordx4"):
            halves[min(n_mfma, 255) // 128].append(("wr", int(re.match(r"buffer_load_dwordx4 v\[(\d+):", ln).group(1))))
    for kt, ops in enumerate(halves):
        rd = {(r - 192) // 32 for kind, r in ops if kind == "rd"}
        wr = {(r - 192) // 32 for kind, r in ops if kind == "wr"}
        assert rd == {kt} and wr == {kt ^ 1}, (kt, rd, wr)


def _replace_all(lines, old, new):
    assert any(ln == old for ln in lines), old
    return [(new if ln == old else ln) for ln in lines]


def _loosen(lines):
    return [re.sub(r"^s_waitcnt lgkmcnt\((\d+)\)$", lambda m: "s_waitcnt lgkmcnt(%d)" % (int(m.group(1)) + 1), ln)
            if ln != "s_waitcnt lgkmcnt(0)" else ln for ln in lines]


MUTATIONS = {
    "vector-memory wait eight too loose (B loads not covered)": lambda L: _replace_all(L, "s_waitcnt vmcnt(8) lgkmcnt(0)", "s_waitcnt vmcnt(16) lgkmcnt(0)"),
    "no barrier": lambda L: [ln for ln in L if ln != "s_barrier"],
    "fragment waits dropped": lambda L: [ln for ln in L if not (ln.startswith("s_waitcnt lgkmcnt(") and ln != "s_waitcnt lgkmcnt(0)")],
    "fragment waits one too loose": _loosen,
    "B k-tile offset never advances": lambda L: [ln for ln in L if ln != "s_add_u32 %[t2], %[t2], 0x8000"],
    "a B load into the set that is being multiplied": lambda L: [ln.replace("buffer_load_dwordx4 v[224:227], %[vb0]", "buffer_load_dwordx4 v[192:195], %[vb0]") for ln in L],
    "A pieces into the slots of the k-tile being read": lambda L: [ln.replace("s_add_u32 %[t4], %[t0], 4", "s_add_u32 %[t4], %[t0], 0") for ln in L],
    "prologue does not wait for B(0) / A(0)": lambda L: _replace_all(L, "s_waitcnt vmcnt(8)", "s_waitcnt vmcnt(24)"),
}


@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_the_harness_sees_seeded_defects_in_the_loop(name):
    pb = H.Problem(7, seed=7)
    ref = reference(pb)
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        try:
            c, _, _ = H.run_p11(pb, lazy_reads, lazy_dma, mutate=MUTATIONS[name])
            e = relerr(c, ref)
            worst = max(worst, e if np.isfinite(e) else 1.0)
        except (RuntimeError, AssertionError):
            worst = 1.0
    assert worst > 1e3 * TOL, (name, worst)


def test_residual_wait_is_load_bearing():
    pb = H.Problem(6, seed=9)
    ref = reference(pb)
    mut = lambda L: _replace_all(L, "s_waitcnt vmcnt(12) lgkmcnt(0)", "s_waitcnt vmcnt(20) lgkmcnt(0)")
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        c, _, _, _ = H.run_p11(pb, lazy_reads, lazy_dma, mutate=mut, res=True)
        e = relerr(c, ref)
        worst = max(worst, e if np.isfinite(e) else 1.0)
    assert worst > 1e3 * TOL
