"""The generated asm loops (alg_amd/csrc/*_loop.inc) are reproducible from their generators, and the e4m3 GEMM schedule -- which
re-reads every operand into the SAME registers while the k-pair is still being multiplied -- is checked as a program: every MFMA
finds the fragments it needs in its operand registers, and its counted wait really covers their reads."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "scripts")
CSRC = os.path.join(ROOT, "alg_amd", "csrc")
GENERATED = {"gen_gemm_p9.py": ("P9_OUT", "gemm_p9_loop.inc"), "gen_gemm_p9_fp8.py": ("P9_FP8_OUT", "gemm_p9_fp8_loop.inc"),
             "gen_attn128_pipe.py": ("ATTN128_PIPE_OUT", "attn128_pipe_loop.inc")}


@pytest.mark.parametrize("script", sorted(GENERATED))
def test_generator_reproduces_the_committed_loop(script, tmp_path):
    var, name = GENERATED[script]
    out = tmp_path / name
    env = dict(os.environ, **{var: str(out)})
    for k in [k for k in env if k.startswith("P9_") and k != var]:
        del env[k]                                   # experiment knobs of gen_gemm_p9.py must not leak in
    r = subprocess.run([sys.executable, os.path.join(SCRIPTS, script)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_bytes() == open(os.path.join(CSRC, name), "rb").read(), "%s is not what %s generates" % (name, script)


def test_fp8_schedule_feeds_every_mfma_the_fragments_it_names():
    sys.path.insert(0, SCRIPTS)
    import gen_gemm_p9_fp8 as F

    for res in (False, True):
        first = [("res%d" % c, F.ktile(True, True, res_copy=c)) for c in range(8)] if res else []
        seq = [("pro", F.prologue())] + first + [("loop", F.ktile(True, True))] * 3 + \
              [("pen", F.ktile(False, True)), ("last", F.ktile(False, False, last=True))]
        texts = F.resolve(seq)                       # asserts that a piece needs the same waits wherever it runs
        holder, issued_at, n_reads, n_mfma = {}, {}, 0, 0
        for name, ins in seq:
            # a new k-tile: what was read as "the next k-tile's k-pair 0" (tag 2) is k-pair 0 now
            holder = {r: ((t[0], t[1], 0, t[3]) if t[2] == 2 else t) for r, t in holder.items()}
            issued_at = {((t[0], t[1], 0, t[3]) if t[2] == 2 else t): i for t, i in issued_at.items()}
            lines = iter(texts[name])
            for kind, info, text in ins:
                if kind == "R":
                    op, i, kp, half = info
                    holder[(op, i, half)] = info     # the 4-register half of operand (op, i) now (eventually) holds this fragment
                    issued_at[info] = n_reads
                    n_reads += 1
                elif kind == "M":
                    n_mfma += 1
                    for tag in info:                 # the MFMA names B8[nt] / A8[mt]: both halves must hold ITS k-pair
                        assert holder.get((tag[0], tag[1], tag[3])) == tag, (name, text, tag, holder.get((tag[0], tag[1], tag[3])))
                    # the wait in front of it (the line before its text) lets at most `w` younger reads stay in flight
                    w = None
                    for ln in lines:
                        if ln.startswith("s_waitcnt lgkmcnt("):
                            w = int(ln[len("s_waitcnt lgkmcnt("):-1])
                        if ln == text:
                            break
                    assert w is not None
                    youngest = max(issued_at[t] for t in info)
                    assert w <= max(0, n_reads - 1 - youngest) or w == 15, (name, text, w, n_reads - 1 - youngest)
        assert n_mfma == 32 * (len(seq) - 1)
