#!/usr/bin/env python3
"""Generates alg_amd/csrc/gemm_p11_loop.inc: the main loop of GEMM schedule 11 as ONE inline-asm statement.

Schedule 11 (round 6) = schedule 10's MFMA shape (v_mfma_f32_16x16x32_bf16) with the WEIGHT operand fetched straight
from L2 into registers instead of through the LDS.  Why (profiles/r6_gemm_p10_loop_ablation_and_variants.txt): with 16-cycle MFMAs the
issue of the sixteen LDS-DMA pieces per k-tile and wave is what schedule 10's loop loses (no LDS-DMA: +39 %; B's pieces as plain loads
into registers nothing waits for: +24 % at the ff2 shape) -- a plain vector load issues far cheaper than an LDS-DMA piece.  Round 5 had
tried that on the 2 x 2 wave layout and lost it again to the duplicate fetch (the two waves that share a column block each need
their own copy).  Here the wave layout is 1 x 4: a wave owns all 256 rows and 64 columns of the tile -- no B fragment is needed by two
waves, A's fragment reads double (32 per k-tile and wave: as many LDS reads as schedules 9 / 10 issue for A and B together), the LDS
ring holds A only (five k-tiles deep), and the LDS-DMA pieces per k-tile and wave halve (8).

B arrives PRE-PACKED in MFMA-fragment order (alg_pack_b_p11; weights are packed once when a model is loaded): per (256-column tile,
64-deep k-tile) 32 KiB = [wave 4][k-step 2][n-block 4][lane 64][16 bytes], so a fragment is ONE coalesced 1 KiB load:
    buffer_load_dwordx4 B(set, ks, bj), vb{(ks*4+bj)>>2}, desc, KB offen offset:((ks*4+bj)&3)*1024
(desc = the wave's 8 KiB slice of the tile's panel, KB = k-tile * 32768 on the scalar unit).

Per k-tile and wave: 128 MFMAs in 32 rows of 4 (row r = the A fragment of k-step r >> 4, m-block r & 15, times the four n-blocks), 32
ds_read_b128 (A, through a four-slot ring THREE rows ahead), 8 B loads for k-tile kt + 1 into the other register set (rows 0-7), 8
LDS-DMA pieces of A(kt + 2) (rows 8-22), the residual form's four quads (rows 24-27), and at the top of row 30 ONE counted wait and ONE
barrier: everything but this k-tile's A pieces (and residual quads) has landed -- B(kt + 1) is in its registers, A(kt + 1) is published
-- then rows 30-31 read the next k-tile's first three A fragments.  The two B sets alternate with the k-tile's parity, so the loop body is a
PAIR of k-tiles and the text carries an odd-count path and both tail parities.  Fragment waits are computed (gen_gemm_p10.place_lgkm_waits).

Accumulators: block (bi 0..15, bj 0..3) = a[16 (2 (bi >> 1) + (bj >> 1)) + 4 (2 (bi & 1) + (bj & 1)) .. + 3]: the 32 x 32 region (mt, nt) of the
wave's 256 x 64 is a[16 (2 mt + nt) ..] (gemm_kernel.h: MT = 8, NT = 2, M16 maps).  C^T layout: MFMA(B fragment, A fragment).
Registers (literal, clobbered): a[0:255]; v[192:223] / v[224:255] B sets 0 / 1 (fragment f = 4 ks + bj at + 4 f); v[176:191] A ring (slot =
row & 3); v[168:175] A DMA offsets per round; v[166:167] A fragment addresses per k-step; v165 residual row offset; v[163:164] are the
set-up's temporaries.  Operands: vl0 / vl1 (lane part of an A fragment address, as schedule 10), vrow / vslot (A DMA lane), vb0 / vb1
(lane * 16 and lane * 16 + 4096: the B loads' vector offsets), pa (64-bit A panel base), db (raw buffer descriptor of the wave's B slice,
4 SGPRs), lda2, rmaxa, nloop (K / 64 - 2), wave1k, t0-t9 scratch SGPRs; residual form: rs, rvoff, ldr16, r0..r31 as schedule 10.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm_p10 import place_lgkm_waits  # noqa: E402

FB = lambda s, f: "v[%d:%d]" % (192 + 32 * s + 4 * f, 192 + 32 * s + 4 * f + 3)
FA = lambda slot: "v[%d:%d]" % (176 + 4 * slot, 176 + 4 * slot + 3)
OFFA = lambda i: "v%d" % (168 + i)
ADA = lambda ks: "v%d" % (166 + ks)
RV = "v165"
TMP0, TMP1 = "v163", "v164"
SLOT = 16384
FIRST_CLOBBERED_VGPR = 163
AHEAD = 3            # rows an A fragment is read ahead of its use (ring of four slots)
BAR = 30             # the k-tile's wait + barrier sit at the top of this row


def acc_index(bi, bj):
    return 16 * (2 * (bi >> 1) + (bj >> 1)) + 4 * (2 * (bi & 1) + (bj & 1))


ACC = lambda bi, bj: "a[%d:%d]" % (acc_index(bi, bj), acc_index(bi, bj) + 3)
# scratch SGPR roles: P ring position of the k-tile being consumed (even slot index), SA its byte base, DA byte base (+ wave1k) of the
# slots that take A(kt + 2), KB byte offset of the k-tile whose B fragments are loaded next, CNT remaining steady-state k-tiles
P, SA, KB, DA, T, T2, PAR, X7, X8, CNT = ("%%[t%d]" % i for i in range(10))

NO_B_WAIT = os.environ.get("P11_NO_B_WAIT") == "1"      # TIMING ONLY: nothing waits for the B loads
NO_DMA = os.environ.get("P11_NO_DMA") == "1"            # TIMING ONLY: no LDS-DMA piece of A is issued in the loop
NO_READS = os.environ.get("P11_NO_READS") == "1"        # TIMING ONLY: no A fragment is read in the loop
NO_B = os.environ.get("P11_NO_B") == "1"                # TIMING ONLY: no B fragment is loaded in the loop
NO_BARRIER = os.environ.get("P11_NO_BARRIER") == "1"    # TIMING ONLY: the per-k-tile workgroup barrier is dropped (the counted wait stays)
# rows (0-27) whose second gap issues one LDS-DMA piece of A(kt + 2) / one B fragment load of k-tile kt + 1
DMA_ROWS = [int(x) for x in os.environ.get("P11_DMA_ROWS", "8,10,12,14,16,18,20,22").split(",")]
B_ROWS = [int(x) for x in os.environ.get("P11_B_ROWS", "0,1,2,3,4,5,6,7").split(",")]
assert len(DMA_ROWS) == 8 and len(B_ROWS) == 8 and max(DMA_ROWS + B_ROWS) < BAR and sorted(DMA_ROWS) == DMA_ROWS and sorted(B_ROWS) == B_ROWS


def read_a(ks, bi, slot):
    return "ds_read_b128 %s, %s offset:%d" % (FA(slot), ADA(ks), bi * 2048)


def load_b(s, f):
    return "buffer_load_dwordx4 %s, %%[vb%d], %%[db], %s offen offset:%d" % (FB(s, f), f >> 2, KB, (f & 3) * 1024)


def dma_a(i):
    off = OFFA(i)
    return ("s_add_u32 m0, %s, %d" % (DA, (i >> 2) * SLOT + (i & 3) * 4096),
            ["global_load_lds_dwordx4 %s, %%[pa]" % off, "v_add_u32 %s, 0x80, %s" % (off, off)])


def slot_math_top():
    """top of a k-tile (P = its ring position, even): where A0 of k-tile kt + 2 goes"""
    return ["s_add_u32 %s, %s, 4" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
            "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (T, T), "s_add_u32 %s, %s, %%[wave1k]" % (DA, T)]


def addr_math():
    out = ["s_lshl_b32 %s, %s, 14" % (SA, P)]
    return out + ["v_add_u32 %s, %s, %%[vl%d]" % (ADA(ks), SA, ks) for ks in range(2)]


def advance():
    """behind the barrier of k-tile kt: P moves on to kt + 1, its fragment addresses"""
    return ["s_add_u32 %s, %s, 2" % (P, P), "s_sub_u32 %s, %s, 10" % (T2, P), "s_cmp_ge_u32 %s, 10" % P,
            "s_cselect_b32 %s, %s, %s" % (P, T2, P)] + addr_math()


RES_COPIES = 8


def res_loads(c):
    """the four residual quads of copy c = the wave's 32-row m-tile c: (half 0, n-tile 0), (0, 1), [row block += 16], (1, 0), (1, 1),
    [+= 16]; quad index it = ((mt * 2 + nt) << 1) | half, as the staged epilogue numbers them (NT = 2)"""
    out = []
    for half in range(2):
        for nt in range(2):
            out.append("buffer_load_dwordx4 %%[r%d], %s, %%[rs], 0 offen offset:%d" % (((c * 2 + nt) << 1) | half, RV, nt * 64))
        out.append("v_add_u32 %s, %%[ldr16], %s" % (RV, RV))
    return out


def ktile(par, b_next, a_dma, barrier, res_copy=None):
    """one k-tile whose B fragments sit in set `par`: 32 rows of 4 MFMAs.  b_next: rows 0-7 load B(kt + 1) into the other set;
    a_dma: rows 8-22 issue A(kt + 2); barrier: the barrier at row 28 and, behind it, the first A fragments of the next k-tile."""
    gaps = [[] for _ in range(128)]
    # A fragment reads, AHEAD = 3 rows ahead: rows 0-28 read rows 3-31 of this k-tile; the barrier sits at the top of row BAR = 30 (the
    # last read of this k-tile, row 31's, was issued in row 28: two rows earlier), behind it rows 30 / 31 read rows 0, 1 / 2 of the next
    for r in range(32 - AHEAD):
        tgt = r + AHEAD
        if not NO_READS:
            gaps[4 * r].append(read_a(tgt >> 4, tgt & 15, tgt & 3))
    if barrier and not NO_READS:
        gaps[4 * 30].append(read_a(0, 0, 0))
        gaps[4 * 30 + 2].append(read_a(0, 1, 1))
        gaps[4 * 31].append(read_a(0, 2, 2))
    if b_next and not NO_B:
        for f in range(8):
            gaps[4 * B_ROWS[f] + (1 if B_ROWS[f] not in DMA_ROWS else 3)].append(load_b(par ^ 1, f))
        gaps[4 * B_ROWS[7] + 2].append("s_add_u32 %s, %s, 0x8000" % (KB, KB))
    if a_dma and not NO_DMA:
        for i in range(8):
            m0, rest = dma_a(i)
            gaps[4 * DMA_ROWS[i] + 1] += [m0, "s_nop 0"] + rest
    if res_copy is not None:
        lds = res_loads(res_copy)     # (load, load, add, load, load, add)
        gaps[4 * 24 + 1] += lds[0:1]
        gaps[4 * 25 + 1] += lds[1:3]
        gaps[4 * 26 + 1] += lds[3:4]
        gaps[4 * 27 + 1] += lds[4:6]
    body = []
    if a_dma:
        body += slot_math_top()
    for j in range(128):
        r, bj = j >> 2, j & 3
        ks, bi = r >> 4, r & 15
        if j == 4 * BAR and barrier:
            # in-order counter: everything issued BEFORE the first piece of A(kt + 2) must have landed
            late_b = sum(1 for f in range(8) if B_ROWS[f] > DMA_ROWS[0]) if (b_next and a_dma) else 0
            assert late_b == 0 or NO_B_WAIT, "a B load behind the first A piece would need the A pieces waited for as well"
            allowed = (8 if a_dma and not NO_DMA else 0) + (4 if res_copy is not None else 0)
            if NO_B_WAIT and b_next and not NO_B:
                allowed += 8
            body += ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % allowed] + ([] if NO_BARRIER else ["s_barrier"]) + advance()
        body.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (ACC(bi, bj), FB(par, 4 * ks + bj), FA(r & 3), ACC(bi, bj)))
        body += gaps[j]
    return body


def setup():
    out = []
    for i in range(8):   # A DMA offsets: row = half * 128 + (i & 3) * 32 + vrow, clamped to the tile's last valid row
        out += ["v_add_u32 %s, 0x%x, %%[vrow]" % (TMP0, i * 32), "v_min_u32 %s, %%[rmaxa], %s" % (TMP0, TMP0),
                "v_mul_lo_u32 %s, %s, %%[lda2]" % (TMP1, TMP0), "v_add_u32 %s, %s, %%[vslot]" % (OFFA(i), TMP1)]
    return out


def prologue():
    out = setup()
    out += ["s_mov_b32 %s, 0" % KB] + [load_b(0, f) for f in range(8)] + ["s_add_u32 %s, %s, 0x8000" % (KB, KB)]
    for kt in range(2):   # A(0) -> slots 0, 1; A(1) -> slots 2, 3
        out += ["s_add_u32 %s, %%[wave1k], %d" % (DA, 2 * kt * SLOT)]
        for i in range(8):
            m0, rest = dma_a(i)
            out += [m0, "s_nop 0"] + rest
    out += ["v_accvgpr_write_b32 a%d, 0" % i for i in range(256)]
    out += ["s_mov_b32 %s, 0" % P] + addr_math()
    out += ["s_waitcnt vmcnt(8)", "s_barrier"]        # B(0) and A(0) have landed (A(1) may be in flight)
    out += [read_a(0, r, r & 3) for r in range(AHEAD) if not NO_READS]
    # the last AHEAD rows of a k-tile read the next k-tile's rows 0 .. AHEAD - 1 in this order: row 28 reads rows (28 + AHEAD - 32 ...)
    return out


def emit(res=False):
    L = prologue()
    entry = {"1", "2", "3"}
    if res:
        entry |= {str(100 + c) for c in range(RES_COPIES + 1)} | {"109"}
        L += ["v_mov_b32 %s, %%[rvoff]" % RV]
    L += ["s_mov_b32 %s, %%[nloop]" % CNT]
    if res:
        # residual form: the first eight steady-state k-tiles each fetch the four quads of one 32-row m-tile; they alternate the B set
        # like every k-tile; short K: the catch-up chain (labels 1xx) fetches what the loop did not get to, then the tails of the
        # parity the next k-tile has (PAR)
        L += ["s_mov_b32 %s, 0" % PAR, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 100f"]
        for c in range(RES_COPIES):
            L += ktile(c & 1, True, True, True, res_copy=c)
            L += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_mov_b32 %s, %d" % (PAR, (c + 1) & 1), "s_cmp_eq_u32 %s, 0" % CNT,
                  "s_cbranch_scc1 %df" % (101 + c)]
    # ---- steady state: pairs of k-tiles ----
    L += ["1:", "s_cmp_lt_u32 %s, 2" % CNT, "s_cbranch_scc1 2f"]
    L += ktile(0, True, True, True) + ktile(1, True, True, True)
    L += ["s_sub_u32 %s, %s, 2" % (CNT, CNT), "s_branch 1b"]
    L += ["2:", "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 3f"]
    L += ktile(0, True, True, True)                                    # an odd steady-state count: one more, then the odd tails
    L += ["4:"] + ktile(1, True, False, True) + ktile(0, False, False, False) + ["s_branch 9f"]
    L += ["3:"] + ktile(0, True, False, True) + ktile(1, False, False, False) + ["s_branch 9f"]
    if res:
        for c in range(RES_COPIES):
            L += ["%d:" % (100 + c)] + res_loads(c)
        L += ["%d:" % (100 + RES_COPIES), "s_cmp_eq_u32 %s, 0" % PAR, "s_cbranch_scc1 3b", "s_branch 4b"]
    L += ["9:", "s_nop 15", "s_nop 15"]
    return place_lgkm_waits(L, entry | {"4"})


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("P11_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "gemm_p11_loop.inc")
    plain, res = emit(), emit(res=True)
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_p11.py -- do not edit.  The main loop of GEMM schedule 11 (1 x 4 waves, B straight into registers) as one asm statement.\n")
        for name, ls in (("ALG_GEMM_P11_LOOP_ASM", plain), ("ALG_GEMM_P11_LOOP_ASM_RES", res)):
            f.write("#define %s \\\n" % name)
            for ln in ls:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
        regs = ["a%d" % i for i in range(256)] + ["v%d" % i for i in range(FIRST_CLOBBERED_VGPR, 256)]
        f.write("#define ALG_GEMM_P11_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
    print("wrote", os.path.normpath(path), len(plain), "+", len(res), "lines,", sum(1 for ln in plain if ln.startswith("v_mfma")), "MFMAs in the plain text")


if __name__ == "__main__":
    main()
