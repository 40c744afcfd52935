#!/usr/bin/env python3
"""Generates alg_amd/csrc/gemm_p10_loop.inc: the main loop of GEMM schedule 10 as ONE inline-asm statement.

Schedule 10 = schedule 9's tile, ring and barrier protocol (gen_gemm_p9.py; gemm_kernel.h has the prose) on the OTHER bf16 MFMA
shape: v_mfma_f32_16x16x32_bf16.  Why (round 6, scripts/micro/mfma_shape.hip, profiles/r6_mfma_shape_power.txt): under the 1400 W
package cap a register-only loop of 16x16x32 sustains 1995-2025 TFLOP/s where the 32x32x16 loop sustains 1817-1828 on the same box
(GEMM-like, 256 accumulators: 1920-1950 against 1790-1805) -- per FLOP the K = 32 form moves half the accumulator words through
the register file, the part clocks 6-10 % higher, and both hot kernels of this build are power-bound.

Per k-tile (64 deep) and wave (128 x 128 of the 256 x 256 tile, one wave per SIMD): 128 MFMAs in 16 ROWS of 8 -- row r multiplies
the A fragment of m-block bi = r & 7, k-step ks = r >> 3 (16 rows x 32 k) into the eight n-blocks -- 32 ds_read_b128 (16 A, 16 B:
as many as schedule 9), 16 LDS-DMA pieces, ONE counted vector-memory wait and ONE barrier.
    B fragments  two sets of eight (one per k-step): v[192:223] / v[224:255].  Set 1 (k-step 1 of this k-tile) is read in rows
                 2-5, set 0 (k-step 0 of the NEXT k-tile) behind the barrier in rows 14-15.
    A fragments  a four-slot ring v[176:191], slot = row & 3, read TWO rows ahead of their use (rows 14 / 15 read rows 0 / 1 of the
                 next k-tile behind the barrier).
    barrier      at the top of row 14: every fragment of this k-tile has been read (the last one, A of row 15, in row 13), so
                 `s_waitcnt vmcnt(8) lgkmcnt(0); s_barrier` publishes k-tile kt + 1 and frees kt's slots, as in schedule 9.
    DMA order    rows 0-3: B(kt + 1) rounds 4-7 (into the slots the previous barrier freed); rows 4-11: A(kt + 2) rounds 0-7; rows
                 14-15: B(kt + 2) rounds 0-3 -- so the eight pieces a barrier may leave in flight are always A(kt + 2).  The
                 prologue stages k-tiles 0 and 1 except B(1) rounds 4-7, which the first k-tile's rows 0-3 issue like every other.
    waits        every `s_waitcnt lgkmcnt(n)` is COMPUTED (place_lgkm_waits): the LDS returns in order, so in front of an MFMA n = the
                 number of reads issued after the youngest one it consumes; the generator checks that the queue of reads in
                 flight is the same at every entry of the loop head and of the tail (prologue, back edge, every residual exit).
Accumulators: block (bi, bj) (16 x 16: m-block bi, n-block bj of the wave's 128 x 128) = a[16 (4 (bi >> 1) + (bj >> 1)) + 4 (2 (bi & 1)
+ (bj & 1)) .. + 3], i.e. the 32 x 32 region (mt, nt) is the same 16 registers schedule 9 uses for it -- the epilogue's
read_acc_block<4 mt + nt>() is unchanged, only the (register, lane) -> (row, column) map differs (gemm_kernel.h, M16).
C^T layout as before: MFMA(B fragment, A fragment), lane = one output row (m = lane & 15), register = 4 consecutive columns.

Register plan inside the statement (named literally, listed as clobbers): a[0:255]; v[192:255] B sets; v[176:191] A ring (v176 /
v177 double as the set-up's temporaries); v[160:167] / v[168:175] DMA byte offsets of the lane into the A / B panel per round,
running in k; v[156:157] / v[158:159] LDS byte address of the A / B fragments per k-step (+ block * 2048 as an immediate); v155
the residual form's running row-block offset.  Operands: as schedule 9 (vl0 / vl1 = lane part of a fragment address for k-step
0 / 1: (lane & 15) * 128 + (((4 ks + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16); vl2 / vl3 are unused).
"""
import os
import re

FB = lambda s, bj: "v[%d:%d]" % (192 + 32 * s + 4 * bj, 192 + 32 * s + 4 * bj + 3)
FA = lambda slot: "v[%d:%d]" % (176 + 4 * slot, 176 + 4 * slot + 3)


def acc_index(bi, bj):
    return 16 * (4 * (bi >> 1) + (bj >> 1)) + 4 * (2 * (bi & 1) + (bj & 1))


ACC = lambda bi, bj: "a[%d:%d]" % (acc_index(bi, bj), acc_index(bi, bj) + 3)
OFFA = lambda i: "v%d" % (160 + i)
OFFB = lambda i: "v%d" % (168 + i)
ADA = lambda ks: "v%d" % (156 + ks)
ADB = lambda ks: "v%d" % (158 + ks)
RV = "v155"
TMP0, TMP1 = "v176", "v177"
SLOT = 16384
FIRST_CLOBBERED_VGPR = 155

# scratch SGPR roles (as schedule 9)
P, SA, SB, DA, T, T2, MA, MB, DB, CNT = ("%%[t%d]" % i for i in range(10))

# experiment knobs.  NO_DMA / NO_READS are timing-only ablations (garbage results); the others are placement variants that stay
# correct (the emulator suite runs them: tests/test_gemm_p10_statement_cpu.py::test_placement_variants_stay_correct)
NO_DMA = os.environ.get("P10_NO_DMA") == "1"
NO_READS = os.environ.get("P10_NO_READS") == "1"
B1_ROWS = [int(x) for x in os.environ.get("P10_B1_ROWS", "2,3,4,5").split(",")]   # rows that read B set 1 (eight reads over these rows)
DMA_GAP = int(os.environ.get("P10_DMA_GAP", "3"))                    # gap of a row (0..7) that carries its LDS-DMA piece
B_EARLY_GAP = int(os.environ.get("P10_B_EARLY_GAP", "6"))
# TIMING ONLY (garbage results): B's eight LDS-DMA pieces per k-tile become eight plain global_load_dwordx4 into registers (the B
# fragment reads stay and stand in for the doubled A reads of a 1 x 4 wave layout): what a weight operand fetched straight into
# registers would cost in issue slots, against the LDS-DMA pieces it replaces
B_DIRECT = os.environ.get("P10_B_DIRECT") == "1"


def read_a(ks, bi, slot):
    return "ds_read_b128 %s, %s offset:%d" % (FA(slot), ADA(ks), bi * 2048)


def read_b(ks, bj, s):
    return "ds_read_b128 %s, %s offset:%d" % (FB(s, bj), ADB(ks), bj * 2048)


def dma(panel, i):
    """(M0 write, [LDS-DMA, offset advance]) of round i (0..7) of a panel; one other instruction has to sit between the M0 write
    and the load that reads it"""
    off = OFFA(i) if panel == "a" else OFFB(i)
    base = "%[pa]" if panel == "a" else "%[pb]"
    dst = DA if panel == "a" else DB
    if B_DIRECT and panel == "b":
        return ("s_nop 0", ["global_load_dwordx4 %s, %s, %s" % (FB(i >> 2, i & 3), off, base), "v_add_u32 %s, 0x80, %s" % (off, off)])
    return ("s_add_u32 m0, %s, %d" % (dst, (i >> 2) * SLOT + (i & 3) * 4096),
            ["global_load_lds_dwordx4 %s, %s" % (off, base), "v_add_u32 %s, 0x80, %s" % (off, off)])


def slot_math_top():
    """top of a k-tile (P = its ring position): where A0 of k-tile kt + 2 goes"""
    return ["s_add_u32 %s, %s, 8" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
            "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (T, T), "s_add_u32 %s, %s, %%[wave1k]" % (DA, T)]


def addr_math():
    """SA / SB and the four fragment address registers from P"""
    out = ["s_add_u32 %s, %s, %%[wm]" % (T, P), "s_lshl_b32 %s, %s, 14" % (SA, T),
           "s_add_u32 %s, %s, %%[wn2]" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
           "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (SB, T)]
    out += ["v_add_u32 %s, %s, %%[vl%d]" % (ADA(ks), SA, ks) for ks in range(2)]
    out += ["v_add_u32 %s, %s, %%[vl%d]" % (ADB(ks), SB, ks) for ks in range(2)]
    return out


def advance(dma_on):
    """behind the barrier of k-tile kt: B0 of kt + 2 takes kt's own first slot; P moves on to kt + 1; its fragment addresses"""
    out = []
    if dma_on:
        out += ["s_lshl_b32 %s, %s, 14" % (T, P), "s_add_u32 %s, %s, %%[wave1k]" % (DB, T)]
    out += ["s_add_u32 %s, %s, 4" % (P, P), "s_sub_u32 %s, %s, 10" % (T2, P), "s_cmp_ge_u32 %s, 10" % P,
            "s_cselect_b32 %s, %s, %s" % (P, T2, P)]
    return out + addr_math()


RES_COPIES = 8        # residual form: the first eight steady-state k-tiles each fetch four of the 32 residual quads


def res_loads(c):
    """the four residual loads of copy c (row block c: m-tile c >> 1, 16-row half c & 1; one per n-tile); quad index
    it = ((mt * 4 + nt) << 1) | half, as the staged epilogue numbers them"""
    mt, half = c >> 1, c & 1
    return ["buffer_load_dwordx4 %%[r%d], %s, %%[rs], 0 offen offset:%d" % (((mt * 4 + nt) << 1) | half, RV, nt * 64)
            for nt in range(4)]


def ktile(b_early, a_dma, barrier, b_late, res_copy=None):
    """one k-tile = 16 rows of 8 MFMAs.  b_early: rows 0-3 issue B(kt + 1) rounds 4-7; a_dma: rows 4-11 A(kt + 2); barrier: the
    barrier at row 14 and, behind it, the first fragments of the next k-tile; b_late: rows 14-15 issue B(kt + 2) rounds 0-3."""
    gaps = [[] for _ in range(128)]      # instructions behind MFMA j

    def put_dma(j, panel, i, filler=None):
        if NO_DMA:
            return
        m0, rest = dma(panel, i)
        gaps[j] += [m0, filler if filler else "s_nop 0"] + rest

    def put_read(j, ins):
        if not NO_READS:
            gaps[j].append(ins)

    for r in range(16):
        # the A fragment of row r + 2 (rows 14 / 15: the next k-tile's rows 0 / 1, behind the barrier)
        if r < 14:
            put_read(8 * r, read_a((r + 2) >> 3, (r + 2) & 7, (r + 2) & 3))
        elif barrier:
            put_read(8 * r, read_a(0, r - 14, (r + 2) & 3))
    nb1 = len(B1_ROWS)
    assert 8 % nb1 == 0 and max(B1_ROWS) < 8      # (set 1 is consumed from row 8 on)
    per = 8 // nb1
    b1_gaps = {1: (2,), 2: (2, 5), 4: (1, 2, 5, 6), 8: (0, 1, 2, 3, 4, 5, 6, 7)}[per]
    for k, r in enumerate(B1_ROWS):           # B set 1 = k-step 1 of this k-tile
        for q in range(per):
            put_read(8 * r + b1_gaps[q], read_b(1, per * k + q, 1))
    if b_early:
        for i in range(4):
            put_dma(8 * i + B_EARLY_GAP, "b", 4 + i)
    if a_dma:
        for i in range(8):
            put_dma(8 * (4 + i) + DMA_GAP, "a", i)
    if res_copy is not None:
        lds = res_loads(res_copy)
        for k, ins in enumerate(lds):
            gaps[8 * (12 + (k >> 1)) + 3 + 3 * (k & 1)].append(ins)
        gaps[8 * 13 + 7].append("v_add_u32 %s, %%[ldr16], %s" % (RV, RV))
    if barrier:                       # B set 0 of the next k-tile: four per row, behind the A fragment
        for r in (14, 15):
            for k in range(4):
                bj = 4 * (r - 14) + k
                put_read(8 * r + 1 + k + (1 if k >= 2 else 0), read_b(0, bj, 0))       # gaps 1, 2, 4, 5
    if b_late:
        for i in range(4):
            r, k = 14 + (i >> 1), i & 1
            put_dma(8 * r + 3 + 3 * k, "b", i)                                          # gaps 3, 6
    body = []
    if a_dma:
        body += slot_math_top()
    for j in range(128):
        r, bj = j >> 3, j & 7
        ks, bi = r >> 3, r & 7
        if j == 8 * 14 and barrier:
            # every fragment of this k-tile has been read; everything but the A(kt + 2) pieces (and this k-tile's residual
            # quads) has landed: publish k-tile kt + 1 / free k-tile kt through ONE barrier
            allowed = ((8 if a_dma else 0) + (4 if res_copy is not None else 0)) if not NO_DMA else 0
            body += ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % allowed, "s_barrier"]
            body += advance(b_late)
        body.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (ACC(bi, bj), FB(ks, bj), FA(r & 3), ACC(bi, bj)))
        body += gaps[j]
    return body


def setup():
    out = []
    # DMA offsets: row r = half * 128 + i * 32 + vrow (vrow = wave * 8 + lane / 8), clamped to the tile's last valid row
    for i in range(8):
        for mx, ld, off in (("%[rmaxa]", "%[lda2]", OFFA(i)), ("%[rmaxb]", "%[ldb2]", OFFB(i))):
            out += ["v_add_u32 %s, 0x%x, %%[vrow]" % (TMP0, i * 32), "v_min_u32 %s, %s, %s" % (TMP0, mx, TMP0),
                    "v_mul_lo_u32 %s, %s, %s" % (TMP1, TMP0, ld), "v_add_u32 %s, %s, %%[vslot]" % (off, TMP1)]
    return out


def prologue():
    out = setup()
    # k-tiles 0 and 1: slots 0-3 and 4-7, order A(0) B(0) A(1) B(1) -- B(1) only rounds 0-3 (rows 0-3 of k-tile 0 issue the rest)
    for kt in range(2):
        out += ["s_add_u32 %s, %%[wave1k], %d" % (DA, (4 * kt) * SLOT), "s_add_u32 %s, %%[wave1k], %d" % (DB, (4 * kt + 2) * SLOT)]
        for panel in ("a", "b"):
            for i in range(8):
                if kt == 1 and panel == "b" and i >= 4:
                    continue
                m0, rest = dma(panel, i)
                out += [m0, "s_nop 0"] + rest
    out += ["v_accvgpr_write_b32 a%d, 0" % i for i in range(256)]
    out += ["s_mov_b32 %s, 0" % P] + addr_math()
    out += ["s_waitcnt vmcnt(12)", "s_barrier"]       # k-tile 0 has landed (A(1) and half of B(1) may be in flight)
    # the first fragments, in the order rows 14-15 of a k-tile issue them
    out += [read_a(0, 0, 0)] + [read_b(0, bj, 0) for bj in range(4)] + [read_a(0, 1, 1)] + [read_b(0, bj, 0) for bj in range(4, 8)]
    return out


_DS = re.compile(r"^ds_read_b128 v\[(\d+):(\d+)\]")
_MF = re.compile(r"^v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\],")


def place_lgkm_waits(lines, entry_labels):
    """Walk the text once (loop bodies appear once), keep the in-order queue of fragment reads, and put the loosest correct
    `s_waitcnt lgkmcnt(n)` in front of every MFMA whose operands are not known to have arrived.  Every label in entry_labels is
    entered from several places: the queue (destination registers in flight, in order) must be the same at all of them."""
    out = []
    queue = []          # destination base registers of the reads not yet known complete, oldest first
    at_label = {}
    def check(label, q):
        if label in at_label:
            assert at_label[label] == q, ("fragment reads in flight differ at label %s" % label, at_label[label], q)
        else:
            at_label[label] = list(q)
    for i, ln in enumerate(lines):
        m = re.match(r"^(\d+):$", ln)
        if m and m.group(1) in entry_labels:
            check(m.group(1), queue)
        b = re.match(r"^s_cbranch_scc[01] (\d+)[bf]$", ln) or re.match(r"^s_branch (\d+)[bf]$", ln)
        if b and b.group(1) in entry_labels:
            check(b.group(1), queue)
        if ln.startswith("s_branch"):
            # what follows is entered through its label only: continue with that label's recorded queue
            nxt = lines[i + 1] if i + 1 < len(lines) else ""
            m2 = re.match(r"^(\d+):$", nxt)
            if m2 and m2.group(1) in at_label:
                queue = list(at_label[m2.group(1)])
        d = _DS.match(ln)
        if d:
            queue.append(int(d.group(1)))
        if "lgkmcnt(0)" in ln:
            queue = []
        f = _MF.match(ln)
        if f:
            need = [int(f.group(1)), int(f.group(2))]
            idx = max((k for k, reg in enumerate(queue) if reg in need), default=-1)
            if idx >= 0:
                n = len(queue) - 1 - idx
                assert n <= 15, n
                out.append("s_waitcnt lgkmcnt(%d)" % n)
                queue = queue[idx + 1:]
        out.append(ln)
    return out


def emit(res=False):
    L = prologue()
    entry = {"1", "2"}
    if not res:
        L += ["s_mov_b32 %s, %%[nloop]" % CNT, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 2f", "1:"]
        L += ktile(True, True, True, True)
        L += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_lg_u32 %s, 0" % CNT, "s_cbranch_scc1 1b", "2:"]
    else:
        # residual form: four quads per k-tile over the first eight steady-state k-tiles; short K: the catch-up chain (labels 1xx)
        # in front of the last two k-tiles fetches what the loop did not get to (as schedule 9)
        entry |= {str(100 + c) for c in range(RES_COPIES + 1)}
        L += ["v_mov_b32 %s, %%[rvoff]" % RV]
        L += ["s_mov_b32 %s, %%[nloop]" % CNT, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 100f"]
        for c in range(RES_COPIES):
            L += ktile(True, True, True, True, res_copy=c)
            L += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 %df" % (101 + c)]
        L += ["1:"]
        L += ktile(True, True, True, True)
        L += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_lg_u32 %s, 0" % CNT, "s_cbranch_scc1 1b", "s_branch 2f"]
        for c in range(RES_COPIES):
            L += ["%d:" % (100 + c)] + res_loads(c) + ["v_add_u32 %s, %%[ldr16], %s" % (RV, RV)]
        L += ["%d:" % (100 + RES_COPIES), "2:"]
    L += ktile(True, False, True, False)      # last but one: B(last) rounds 4-7, then the barrier drains everything
    L += ktile(False, False, False, False)    # last
    L += ["s_nop 15", "s_nop 15"]             # the last MFMAs' results before any v_accvgpr_read of the epilogue
    if NO_READS:
        return L
    return place_lgkm_waits(L, entry)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("P10_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "gemm_p10_loop.inc")
    plain, res = emit(), emit(res=True)
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_p10.py -- do not edit.  The main loop of GEMM schedule 10 (16x16x32 MFMAs) as one asm statement.\n")
        for name, ls in (("ALG_GEMM_P10_LOOP_ASM", plain), ("ALG_GEMM_P10_LOOP_ASM_RES", res)):
            f.write("#define %s \\\n" % name)
            for ln in ls:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
        regs = ["a%d" % i for i in range(256)] + ["v%d" % i for i in range(FIRST_CLOBBERED_VGPR, 256)]
        f.write("#define ALG_GEMM_P10_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
    n_mfma = sum(1 for ln in plain if ln.startswith("v_mfma"))
    print("wrote", os.path.normpath(path), len(plain), "+", len(res), "lines,", n_mfma, "MFMAs in the plain text,",
          sum(1 for ln in plain if ln.startswith("s_waitcnt lgkmcnt")), "computed fragment waits")


if __name__ == "__main__":
    main()
