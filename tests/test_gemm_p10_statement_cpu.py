"""GEMM schedule 10's generated main-loop statements (scripts/gen_gemm_p10.py -> alg_amd/csrc/gemm_p10_loop.inc: schedule 9's tile,
ring and barrier protocol on v_mfma_f32_16x16x32_bf16 -- the MFMA shape that sustains ~10 % more under the package power cap)
checked AS PROGRAMS on the CPU like schedule 9's (test_gemm_p9_statement_cpu.py): four waves of one 256 x 256 tile in the
instruction-level emulator, fragment reads and LDS-DMA / buffer loads landing only at the counted wait that covers them, the
accumulators (16 x 16 blocks in schedule 9's 32 x 32 register regions) against a float64 A B^T, the residual tile bit for bit, and
seeded defects that the harness has to catch.  The fragment waits of this statement are COMPUTED by the generator; the test that
loosens every one of them by one shows they are tight.  No GPU."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gemm_emu as H  # noqa: E402
import gen_gemm_p10 as G10  # noqa: E402

MODES = [(True, False), (False, True), (True, True)]     # (lazy fragment reads, lazy DMA / buffer loads)
TOL = 2e-6


def relerr(c, ref):
    return float(np.abs(c - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("nk", [2, 3, 4, 7, 12])
def test_plain_statement_computes_the_tile_under_the_weakest_memory_ordering(nk):
    pb = H.Problem(nk, seed=nk)
    ref = pb.reference()
    for lazy_reads, lazy_dma in MODES:
        c, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, sched=10)
        assert relerr(c, ref) < TOL, (nk, lazy_reads, lazy_dma, relerr(c, ref))


def test_plain_statement_on_a_ragged_tile_never_reads_outside_the_panels():
    pb = H.Problem(6, seed=21, rows_a=200, rows_b=130)
    c, _, _ = H.run_plain(pb, True, True, sched=10)
    assert np.isfinite(c).all() and relerr(c, pb.reference()) < TOL


@pytest.mark.parametrize("nk", [2, 3, 5, 9, 10, 11, 13])
def test_residual_statement_returns_the_tile_and_the_residual(nk):
    rows = 256 if nk % 2 else 216
    pb = H.Problem(nk, seed=100 + nk, rows_a=rows)
    ref = pb.reference()
    rref = np.zeros((256, 256))
    rref[:rows] = pb.r
    for lazy_reads, lazy_dma in (MODES if nk in (2, 10) else MODES[2:]):
        c, r, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, res=True, sched=10)
        assert relerr(c, ref) < TOL, (nk, relerr(c, ref))
        assert np.array_equal(r, rref), nk


def test_the_text_in_the_tree_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "p10.inc"
    os.environ["P10_OUT"] = str(out)
    try:
        G10.main()
    finally:
        del os.environ["P10_OUT"]
    with open(os.path.join(ROOT, "alg_amd", "csrc", "gemm_p10_loop.inc")) as f:
        assert f.read() == out.read_text()


def test_statement_shape_per_k_tile():
    """per steady-state k-tile and wave: 128 MFMAs of 16x16x32 (= 64 of 32x32x16), 32 fragment reads, 16 LDS-DMA pieces, one barrier"""
    L = G10.emit()
    body = L[L.index("1:"):L.index("s_cbranch_scc1 1b")]
    count = lambda pre: sum(1 for ln in body if ln.startswith(pre))
    assert count("v_mfma_f32_16x16x32_bf16") == 128 and count("ds_read_b128") == 32
    assert count("global_load_lds_dwordx4") == 16 and count("s_barrier") == 1
    # every accumulator register is written by exactly two MFMAs per k-tile (one per k-step), 256 registers in all
    acc = [re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\]", ln).groups() for ln in body if ln.startswith("v_mfma")]
    assert sorted(int(a) for a, _ in acc) == sorted(2 * list(range(0, 256, 4)))
    # an M0 write never directly precedes the LDS-DMA that reads it (one wait state)
    for a, b in zip(body, body[1:]):
        assert not (a.startswith("s_add_u32 m0") and b.startswith("global_load_lds"))


def _replace_all(lines, old, new):
    assert any(ln == old for ln in lines), old
    return [(new if ln == old else ln) for ln in lines]


def _loosen_fragment_waits(lines):
    """every computed fragment wait one looser: the youngest fragment an MFMA consumes may still be in flight"""
    out = []
    for ln in lines:
        m = re.match(r"^s_waitcnt lgkmcnt\((\d+)\)$", ln)
        out.append("s_waitcnt lgkmcnt(%d)" % (int(m.group(1)) + 1) if m and int(m.group(1)) > 0 or (m and ln != "s_waitcnt lgkmcnt(0)") else ln)
    return out


MUTATIONS = {
    "DMA wait four too loose": lambda L: _replace_all(L, "s_waitcnt vmcnt(8) lgkmcnt(0)", "s_waitcnt vmcnt(12) lgkmcnt(0)"),
    "no barrier": lambda L: [ln for ln in L if ln != "s_barrier"],
    "fragment waits dropped": lambda L: [ln for ln in L if not (ln.startswith("s_waitcnt lgkmcnt(") and ln != "s_waitcnt lgkmcnt(0)")],
    "fragment waits one too loose": _loosen_fragment_waits,
    "late B DMA into the wrong slot": lambda L: [ln.replace("s_add_u32 m0, %[t8], 4096", "s_add_u32 m0, %[t8], 20480") for ln in L],
    "early B DMA into the wrong half-tile": lambda L: [ln.replace("s_add_u32 m0, %[t8], 16384", "s_add_u32 m0, %[t8], 0") for ln in L],
    "prologue does not wait for the first k-tile": lambda L: _replace_all(L, "s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(28)"),
    "barrier one row early": lambda L: _barrier_one_row_early(L),
}


def _barrier_one_row_early(lines):
    """the steady-state barrier (and its waits) in front of row 13 instead of row 14: the A fragment of row 15 is then read from a
    slot the B pieces of k-tile kt + 2 may already be landing in"""
    out = list(lines)
    start = out.index("1:")
    i = next(k for k in range(start, len(out)) if out[k] == "s_waitcnt vmcnt(8) lgkmcnt(0)")
    j = i
    while not out[j].startswith("v_mfma"):      # wait, barrier, slot math of the next k-tile
        j += 1
    block = out[i:j]
    del out[i:j]
    # back over eight MFMAs (one row)
    k, seen = i - 1, 0
    while seen < 8:
        if out[k].startswith("v_mfma"):
            seen += 1
        k -= 1
    out[k + 1:k + 1] = block
    return out


@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_the_harness_sees_seeded_defects_in_the_loop(name):
    pb = H.Problem(7, seed=7)
    ref = pb.reference()
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        try:
            c, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, mutate=MUTATIONS[name], sched=10)
            e = relerr(c, ref)
            worst = max(worst, e if np.isfinite(e) else 1.0)
        except (RuntimeError, AssertionError):          # deadlock / runaway / memory fault: also a detection
            worst = 1.0
    assert worst > 1e3 * TOL, (name, worst)


def test_residual_wait_is_load_bearing():
    pb = H.Problem(6, seed=9)
    ref = pb.reference()
    mut = lambda L: _replace_all(L, "s_waitcnt vmcnt(12) lgkmcnt(0)", "s_waitcnt vmcnt(16) lgkmcnt(0)")
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        c, _, _, _ = H.run_plain(pb, lazy_reads, lazy_dma, mutate=mut, res=True, sched=10)
        e = relerr(c, ref)
        worst = max(worst, e if np.isfinite(e) else 1.0)
    assert worst > 1e3 * TOL


@pytest.mark.parametrize("env", [{"P10_B1_ROWS": "0,1,2,3"}, {"P10_B1_ROWS": "0,1,2,3,4,5,6,7"}, {"P10_DMA_GAP": "6"},
                                 {"P10_B1_ROWS": "1,3,5,7", "P10_DMA_GAP": "7", "P10_B_EARLY_GAP": "4"}])
def test_placement_variants_stay_correct(env, monkeypatch):
    """the generator's placement knobs (which rows read B set 1, which gap of a row carries its LDS-DMA piece) move instructions, never
    the protocol: the computed waits follow, and every variant still computes the tile under lazy reads + lazy DMA"""
    import importlib
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    importlib.reload(G10)
    try:
        H.GP10 = G10
        for nk, res in ((5, False), (9, True)):
            pb = H.Problem(nk, seed=40 + nk)
            out = H.run_plain(pb, True, True, res=res, sched=10)
            assert relerr(out[0], pb.reference()) < TOL, (env, nk, res)
    finally:
        for k in env:
            monkeypatch.delenv(k)
        importlib.reload(G10)
        H.GP10 = G10
