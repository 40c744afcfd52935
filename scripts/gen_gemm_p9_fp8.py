#!/usr/bin/env python3
"""Generates alg_amd/csrc/gemm_p9_fp8_loop.inc: schedule 9's main loop for OCP e4m3 operands (alg_gemm_fp8, BASELINE config 5).

Same tile, ring, DMA and barrier protocol as gen_gemm_p9.py (read its header first): a k-tile is again 128 bytes per row -- 128
fp8 values instead of 64 bf16 -- so the sixteen LDS-DMA instructions and the thirty-two 16-byte fragment reads per k-tile and wave
are byte for byte the same.  What changes is the matrix instruction: v_mfma_scale_f32_32x32x64_f8f6f4 (all block scales 2^0: a
plain K = 64 e4m3 contraction at twice the bf16 rate, 16 passes) takes 32 bytes per lane and operand = the fragments of two
k-steps in EIGHT CONSECUTIVE registers.  32 MFMAs per k-tile (two k-pairs kp x 4 x 4 blocks) of twice the length.

Register plan: the same 64 fragment registers as the bf16 loop, but as ONE set of eight 8-register operands
    B8[nt] = v[192 + 8 nt .. + 7]   A8[mt] = v[224 + 8 mt .. + 7]       (low four: k-step 2 kp, high four: k-step 2 kp + 1)
There is no second set: an operand is re-read for the next k-pair as soon as its last MFMA of this k-pair has been issued
(A8[mt] behind MFMA (mt, 3), B8[nt] behind MFMA (3, nt)) and is first used three or more 64-cycle MFMAs later.  The barrier of
a k-tile sits in front of MFMA 18: by then every fragment read of the k-tile has been issued for two MFMAs (lgkmcnt(0) is
free), and the reads that follow (from gap 19 on) belong to the next k-tile.  v163 holds the E8M0 scales (0x7f7f7f7f).
Everything else -- operands, scratch SGPR roles, slot arithmetic, residual fetch inside the loop -- is gen_gemm_p9.py's.
"""
import os

import gen_gemm_p9 as G

FB8 = lambda nt: "v[%d:%d]" % (192 + 8 * nt, 192 + 8 * nt + 7)
FA8 = lambda mt: "v[%d:%d]" % (224 + 8 * mt, 224 + 8 * mt + 7)
SC = "v163"
BARRIER_AT = 18


def read(op, i, ks):
    """16-byte fragment of k-step ks (0..3) of B[nt = i] / A[mt = i] into its half of the 8-register operand"""
    base = (192 if op == "b" else 224) + 8 * i + 4 * (ks & 1)
    return ("R", (op, i, ks >> 1, ks & 1), "ds_read_b128 v[%d:%d], %s offset:%d" % (base, base + 3, G.ADB(ks) if op == "b" else G.ADA(ks),
                                                                                      i * 4096))


def operand_reads(op, i, kp):
    return [read(op, i, 2 * kp), read(op, i, 2 * kp + 1)]


def mfma(mt, nt, kp):
    need = [("b", nt, kp, 0), ("b", nt, kp, 1), ("a", mt, kp, 0), ("a", mt, kp, 1)]
    return ("M", need, "v_mfma_scale_f32_32x32x64_f8f6f4 %s, %s, %s, %s, %s, %s op_sel_hi:[0,0,0]"
            % (G.ACC(mt, nt), FB8(nt), FA8(mt), G.ACC(mt, nt), SC, SC))


def ktile(dma_on, barrier_on, res_copy=None, last=False):
    """one k-tile as a list of ("R" | "M" | "I", info, text): 32 MFMAs, the instructions of gap j right behind MFMA j"""
    m0s = [[] for _ in range(32)]
    rds = [[] for _ in range(32)]
    posts = [[] for _ in range(32)]

    def put_dma(j, panel, i):
        m0, rest = G.dma(panel, i)
        m0s[j].append(m0)
        posts[j] += rest

    pre = []
    if dma_on:
        pre += G.slot_math_top()
        for i in range(8):
            put_dma(2 * i + 1, "a", i)                    # A(kt + 2): gaps 1, 3, ..., 15
        if barrier_on:
            for i, j in enumerate((18, 20, 21, 22, 24, 25, 26, 29)):
                put_dma(j, "b", i)                        # B(kt + 2): behind the barrier, into k-tile kt's first slots
    # k-pair 1 of THIS k-tile, read during k-pair 0
    for mt in range(4):
        rds[4 * mt + 3] += operand_reads("a", mt, 1)
    for nt in range(4):
        rds[12 + nt] += operand_reads("b", nt, 1)
    # k-pair 0 of the NEXT k-tile, read during k-pair 1 (behind the barrier: the fragment addresses are the next k-tile's by then)
    if not last:
        for mt in range(4):
            rds[16 + 4 * mt + 3] += [("R", ("a", mt, 2, h), t) for (_, (_, _, _, h), t) in operand_reads("a", mt, 0)]
        for nt in range(4):
            rds[28 + nt] += [("R", ("b", nt, 2, h), t) for (_, (_, _, _, h), t) in operand_reads("b", nt, 0)]
    if res_copy is not None:
        for g, ins in zip((2, 6, 10, 14), G.res_loads(res_copy)):
            posts[g].append(ins)
        posts[14].append("v_add_u32 %s, %%[ldr16], %s" % (G.RV, G.RV))
    out = [("I", None, t) for t in pre]
    for j in range(32):
        kp, q = j >> 4, j & 15
        if j == BARRIER_AT and barrier_on:
            out.append(("I", None, "s_waitcnt vmcnt(%d) lgkmcnt(0)" % ((12 if res_copy is not None else 8) if dma_on else 0)))
            out.append(("B", None, "s_barrier"))
            out += [("I", None, t) for t in (G.slot_math_advance() if dma_on else G.advance_no_dma())]
        out.append(mfma(q >> 2, q & 3, kp))
        out += [("I", None, t) for t in m0s[j]]
        out += rds[j]
        if m0s[j] and not rds[j]:
            out.append(("I", None, "s_nop 0"))
        out += [("I", None, t) for t in posts[j]]
    return out


def resolve(pieces):
    """pieces: list of (name, instruction list) in execution order.  LDS returns in order: in front of every MFMA a counted wait
    = the number of reads issued after the youngest fragment it needs.  Operand tags carry k-pair 0 / 1 of the current k-tile;
    reads tagged k-pair 2 are the next k-tile's k-pair 0.  Returns {name: [text lines]} (the LAST occurrence of a name wins,
    i.e. the steady-state form; every occurrence is checked to need the same waits)."""
    issued = {}          # tag -> running index of the youngest read with that tag
    n = 0
    texts = {}
    for name, ins in pieces:
        lines = []
        # a new k-tile: what was read as "k-pair 2" is now k-pair 0
        for tag in [t for t in issued if t[2] == 2]:
            issued[(tag[0], tag[1], 0, tag[3])] = issued.pop(tag)
        for kind, info, text in ins:
            if kind == "R":
                issued[info] = n
                n += 1
            elif kind == "B":
                pass
            elif kind == "M":
                idx = max(issued[t] for t in info)
                lines.append("s_waitcnt lgkmcnt(%d)" % min(15, n - 1 - idx))
            lines.append(text)
        if name in texts:
            assert texts[name] == lines, "piece %s needs different waits in different contexts" % name
        texts[name] = lines
    return texts


def prologue():
    out = [("I", None, t) for t in G.setup()]
    for kt in range(2):
        out += [("I", None, t) for t in ("s_add_u32 %s, %%[wave1k], %d" % (G.DA, (4 * kt) * G.SLOT),
                                          "s_add_u32 %s, %%[wave1k], %d" % (G.DB, (4 * kt + 2) * G.SLOT))]
        for panel in ("a", "b"):
            for i in range(8):
                m0, rest = G.dma(panel, i)
                out += [("I", None, t) for t in [m0, "s_nop 0"] + rest]
    out += [("I", None, "v_accvgpr_write_b32 a%d, 0" % i) for i in range(256)]
    out += [("I", None, "v_mov_b32 %s, 0x7f7f7f7f" % SC)]
    out += [("I", None, t) for t in ["s_mov_b32 %s, 0" % G.P] + G.addr_math() + ["s_waitcnt vmcnt(16)", "s_barrier"]]
    # k-pair 0 of k-tile 0, in the order the loop re-reads it (A8[0..2], B8[0..2], A8[3], B8[3])
    order = [("a", 0), ("a", 1), ("a", 2), ("b", 0), ("b", 1), ("b", 2), ("a", 3), ("b", 3)]
    for op, i in order:
        out += operand_reads(op, i, 0)
    return out


def emit(res):
    I = lambda ts: [("I", None, t) for t in ts]
    if not res:
        seq = [("pro", prologue()), ("loop", ktile(True, True)), ("loop", ktile(True, True)),
               ("pen", ktile(False, True)), ("last", ktile(False, False, last=True))]
        t = resolve(seq)
        # the penultimate k-tile directly behind the prologue (K = 256) must need the same waits
        t2 = resolve([("pro", prologue()), ("pen", ktile(False, True)), ("last", ktile(False, False, last=True))])
        assert t2["pen"] == t["pen"] and t2["last"] == t["last"]
        lines = t["pro"] + ["s_mov_b32 %s, %%[nloop]" % G.CNT, "s_cmp_eq_u32 %s, 0" % G.CNT, "s_cbranch_scc1 2f", "1:"] + t["loop"] + \
            ["s_sub_u32 %s, %s, 1" % (G.CNT, G.CNT), "s_cmp_lg_u32 %s, 0" % G.CNT, "s_cbranch_scc1 1b", "2:"] + t["pen"] + t["last"]
    else:
        seq = [("pro", prologue())] + [("res%d" % c, ktile(True, True, res_copy=c)) for c in range(G.RES_COPIES)] + \
              [("loop", ktile(True, True)), ("loop", ktile(True, True)), ("pen", ktile(False, True)),
               ("last", ktile(False, False, last=True))]
        t = resolve(seq)
        lines = t["pro"] + ["v_mov_b32 %s, %%[rvoff]" % G.RV, "s_mov_b32 %s, %%[nloop]" % G.CNT, "s_cmp_eq_u32 %s, 0" % G.CNT,
                            "s_cbranch_scc1 100f"]
        for c in range(G.RES_COPIES):
            lines += t["res%d" % c] + ["s_sub_u32 %s, %s, 1" % (G.CNT, G.CNT), "s_cmp_eq_u32 %s, 0" % G.CNT,
                                       "s_cbranch_scc1 %df" % (101 + c)]
        lines += ["1:"] + t["loop"] + ["s_sub_u32 %s, %s, 1" % (G.CNT, G.CNT), "s_cmp_lg_u32 %s, 0" % G.CNT, "s_cbranch_scc1 1b",
                                       "s_branch 2f"]
        for c in range(G.RES_COPIES):
            lines += ["%d:" % (100 + c)] + G.res_loads(c) + ["v_add_u32 %s, %%[ldr16], %s" % (G.RV, G.RV)]
        lines += ["%d:" % (100 + G.RES_COPIES), "2:"] + t["pen"] + t["last"]
    return lines + ["s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]   # 16-pass MFMAs: results before any v_accvgpr_read


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("P9_FP8_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "gemm_p9_fp8_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_p9_fp8.py -- do not edit.  GEMM schedule 9, e4m3 operands: the main loop as one asm statement.\n")
        for name, ls in (("ALG_GEMM_P9_FP8_LOOP_ASM", emit(False)), ("ALG_GEMM_P9_FP8_LOOP_ASM_RES", emit(True))):
            f.write("#define %s \\\n" % name)
            for ln in ls:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
    print("wrote", os.path.normpath(path), len(emit(False)), "lines,", sum(1 for l in emit(False) if l.startswith("v_mfma")), "MFMAs")


if __name__ == "__main__":
    main()
