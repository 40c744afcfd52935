"""The d = 64 attention statement on v_mfma_f32_16x16x32_bf16 (scripts/gen_attn_m16.py -> alg_amd/csrc/attn64_m16_loop.inc; round 6:
the MFMA shape that sustains ~10 % more under the package power cap) checked AS A PROGRAM on the CPU like the other single-statement
loops (test_attn_q64_statement_cpu.py): eight waves x 32 queries of one 256-query unit in the instruction-level emulator, fragment
reads and LDS-DMA landing only at the counted wait that covers them, against float64 attention; the K-fragment row map that makes
the S^T blocks of a lane the PV operand in V^T's stored order, the kernel's own K-tile chunk XOR, the two row sums per lane and the
refusal (exit code 1) on a row sum >= 2^80 are all part of what runs.  The fragment waits are COMPUTED by the generator; seeded
defects show that they -- and the DMA wait and the barrier -- are load-bearing.  No GPU."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import attn_emu as H  # noqa: E402
import gen_attn_m16 as GM  # noqa: E402

MODES = [(True, False), (False, True), (True, True)]
TOL = 6e-3      # bf16 probabilities: 2^-9 relative per term; measured 1.7-2.1e-3 on these problems (the 32x32x16 statement: the same)


def relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("T,seed", [(15, 11), (20, 12), (9, 13)])
def test_m16_statement_computes_attention_under_the_weakest_memory_ordering(T, seed):
    pb = H.Problem(64, T, seed=seed, prescaled=True)
    ref = pb.reference()
    for lazy_reads, lazy_dma in MODES:
        out, t_exit, codes, _, _ = H.run_m16_statement(pb, lazy_reads, lazy_dma)
        assert t_exit == 1 + 4 * ((T - 3 - 1) // 4) and codes == [0] * 8          # whole groups of four while t + 4 <= T - 3
        assert relerr(out, ref) < TOL, (lazy_reads, lazy_dma, relerr(out, ref))


def test_m16_statement_re_entered_at_a_later_t_with_strided_panels():
    """the frame re-enters the statement at any t = 1 (mod 4) behind a refused tile; K and Q rows of a [S, 3, D] tensor"""
    pb = H.Problem(64, 18, seed=3, q_rs=192, k_rs=192, prescaled=True)
    out, t_exit, codes, _, _ = H.run_m16_statement(pb, True, True, t0=5)
    assert t_exit == 13 and codes == [0] * 8
    assert relerr(out, pb.reference()) < TOL


def test_m16_statement_leaves_with_code_1_when_a_row_sum_passes_2_to_the_80():
    """keys of tile 3 scaled so that every query's scores there are ~ +-100 log2 units and some row sums pass 2^80: every wave that
    sees one leaves in iteration 3 with code 1, its l and O untouched by that tile (the harness redoes the tile in float64)"""
    pb = H.Problem(64, 15, seed=21, prescaled=True)
    pb.k[3 * 64:4 * 64] = H.bf16_round(pb.k[3 * 64:4 * 64] * 96.0)
    pb.pack()
    s = pb.q.astype(np.float64) @ pb.k.astype(np.float64).T
    assert (np.exp2(s[:, 192:256]).sum(axis=1) > 2.0 ** 80).any()
    _, t_exit, codes, _, _ = H.run_m16_statement(pb, True, True)
    assert codes[0] == 1 and t_exit == 3, (codes, t_exit)


def test_the_text_in_the_tree_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "m16.inc"
    os.environ["ATTN_M16_OUT"] = str(out)
    try:
        GM.main()
    finally:
        del os.environ["ATTN_M16_OUT"]
    with open(os.path.join(ROOT, "alg_amd", "csrc", "attn64_m16_loop.inc")) as f:
        assert f.read() == out.read_text()


def _loop_body(lines):
    return lines[lines.index("11:"):lines.index("s_cbranch_scc1 11b")]


def test_statement_shape_and_static_hazard_rules():
    L = GM.emit()
    body = _loop_body(L)
    count = lambda pre: sum(1 for ln in body if ln.startswith(pre))
    # four iterations per trip: 32 MFMAs, 16 fragment reads, 2 DMA pieces, 32 exps, 16 packs, one barrier each
    assert count("v_mfma_f32_16x16x32_bf16") == 128 and count("ds_read_b128") == 64 and count("global_load_lds_dwordx4") == 8
    assert count("v_exp_f32") == 128 and count("v_cvt_pk_bf16_f32") == 64 and count("s_barrier") == 4
    # at most four VALU / memory instructions between two MFMAs of an iteration's stream (the MFMA's 16-cycle shadow)
    run, worst = 0, 0
    for ln in body:
        if ln.startswith("v_mfma"):
            worst, run = max(worst, run), 0
        elif not ln.startswith(("s_", "11:", "12:")):
            run += 1
    assert worst <= 14        # (the iteration boundary: row-sum check, next pre-exps)
    mf = [i for i, ln in enumerate(L) if ln.startswith("v_mfma")]
    for a, b in zip(L, L[1:]):        # M0 write -> LDS-DMA needs one instruction in between; exp -> its first use likewise
        assert not (a.startswith("s_add_u32 m0") and b.startswith("global_load_lds"))
        if a.startswith("v_exp_f32"):
            dst = a.split()[1].rstrip(",")
            assert dst not in re.findall(r"v\d+", b.split(None, 1)[1] if " " in b else ""), (a, b)
    # XDL write -> VALU read: a score register is read by an exp at least three MFMAs after the MFMA that completed it
    last_write = {}
    for i, ln in enumerate(L):
        m = re.match(r"v_mfma_f32_16x16x32_bf16 v\[(\d+):(\d+)\]", ln)
        if m:
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                last_write[r] = i
        e = re.match(r"v_exp_f32 v\d+, v(\d+)$", ln)
        if e and int(e.group(1)) in last_write:
            w = last_write[int(e.group(1))]
            between = sum(1 for k in mf if w < k < i) + sum(16 for ln2 in L[w:i] if ln2 == "s_nop 15")
            assert between >= 3, (ln, w, i, between)


def _replace_all(lines, old, new):
    assert any(ln == old for ln in lines), old
    return [(new if ln == old else ln) for ln in lines]


def _loosen(lines):
    return [re.sub(r"^s_waitcnt lgkmcnt\((\d+)\)$", lambda m: "s_waitcnt lgkmcnt(%d)" % (int(m.group(1)) + 1), ln)
            if ln != "s_waitcnt lgkmcnt(0)" else ln for ln in lines]


MUTATIONS = {
    "fragment waits one too loose": _loosen,
    "fragment waits dropped": lambda L: [ln for ln in L if not (ln.startswith("s_waitcnt lgkmcnt(") and ln != "s_waitcnt lgkmcnt(0)")],
    "DMA wait two too loose": lambda L: _replace_all(L, "s_waitcnt vmcnt(2)", "s_waitcnt vmcnt(4)"),
    "no barrier": lambda L: [ln for ln in L if ln != "s_barrier"],
    "K piece into the wrong ring slot": lambda L: [ln.replace("s_add_u32 m0, %[wk], 8192", "s_add_u32 m0, %[wk], 16384") for ln in L],
    "K fragment of the wrong block": lambda L: [ln.replace("%[lk1] offset:13312", "%[lk1] offset:12288") for ln in L],
    "row sum of query block 1 added to block 0": lambda L: _replace_all(L, "v_add_f32 %[l1], %[l1], v123", "v_add_f32 %[l0], %[l0], v123"),
}


@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_the_harness_sees_seeded_defects(name):
    pb = H.Problem(64, 15, seed=11, prescaled=True)
    ref = pb.reference()
    worst = 0.0
    for lazy_reads, lazy_dma in MODES:
        try:
            out, _, _, _, _ = H.run_m16_statement(pb, lazy_reads, lazy_dma, mutate=MUTATIONS[name])
            e = relerr(out, ref)
            worst = max(worst, e if np.isfinite(e) else 1.0)
        except (RuntimeError, AssertionError):
            worst = 1.0
    assert worst > 10 * TOL, (name, worst)
