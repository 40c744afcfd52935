#!/usr/bin/env python3
"""Generates alg_amd/csrc/gemm_p9_loop.inc: the hand-scheduled main loop of GEMM schedule 9 as ONE inline-asm statement.

Schedule 9 (see gemm_p9.hip for the prose): 4 waves x 128x128 of a 256x256 tile, one wave per SIMD, 256 fp32 accumulators
in a[0:255] for the whole tile; K streamed in 64-deep k-tiles as four 16 KiB half-tiles (A rows 0-127, A rows 128-255,
B rows 0-127, B rows 128-255: whole 128-byte lines) through a TEN-slot LDS ring (160 KiB = 2.5 k-tiles); per k-tile and wave
64 v_mfma_f32_32x32x16_bf16, 32 ds_read_b128, 16 global_load_lds_dwordx4, ONE counted wait (vmcnt(8)) and ONE barrier.
Nothing in the loop is compiler-scheduled: every memory instruction sits at a fixed place between two MFMAs.

Register plan inside the statement (all named literally and listed as clobbers by the caller):
    a[0:255]     accumulators, block (mt, nt) = a[16 (4 mt + nt) .. + 15]  (C^T layout: MFMA(B fragment, A fragment))
    v[192:223]   fragment set 0: B[nt] = v[192 + 4 nt ..], A[mt] = v[208 + 4 mt ..]
    v[224:255]   fragment set 1
    v[176:183]   DMA byte offsets of this lane into the A panel (rounds i = 0..7: half-tile i >> 2, round i & 3), running in k
    v[184:191]   the same for the B panel
    v[168:171]   LDS byte addresses of the A fragments of the current k-tile, one per k-step (+ mt * 4096 as an immediate)
    v[172:175]   ... of the B fragments (+ nt * 4096)
    v[160:161]   temporaries of the set-up (v[160:167] are reserved)
Operands (named): vl0-vl3 lane part of a fragment address per k-step; vrow, vslot: row / byte-in-row this lane fetches;
pa, pb 64-bit panel bases; lda2, ldb2 row pitches in bytes; rmaxa, rmaxb last valid row of the tile; nloop steady-state
trips (K / 64 - 2); wm, wn wave coordinates; wave1k = lds base + wave * 1024; t0-t9 scratch SGPRs.
"""
import os

# experiment knobs (round 3-5 timing ablations, docs/lab_notebook_r5.md item 20): they produce garbage results
NO_DMA = os.environ.get("P9_NO_DMA") == "1"          # no LDS-DMA inside the loop (the prologue's two k-tiles are re-read)
NO_READS = os.environ.get("P9_NO_READS") == "1"      # no fragment reads inside the loop
NO_BARRIER = os.environ.get("P9_NO_BARRIER") == "1"  # no s_barrier inside the loop
B_FAST = os.environ.get("P9_B_FAST") == "1"          # B(kt + 2): one DMA per MFMA gap right behind the barrier (more slack)
A_GAPS = os.environ.get("P9_A_GAPS", "")             # comma list of the eight gaps (0..47) that carry the A DMAs
BUF = os.environ.get("P9_BUF") == "1"                # DMA as buffer_load ... lds: k advances in the scalar offset (no v_add per DMA)

FB = lambda s, nt: "v[%d:%d]" % (192 + 32 * s + 4 * nt, 192 + 32 * s + 4 * nt + 3)
FA = lambda s, mt: "v[%d:%d]" % (208 + 32 * s + 4 * mt, 208 + 32 * s + 4 * mt + 3)
ACC = lambda mt, nt: "a[%d:%d]" % (16 * (4 * mt + nt), 16 * (4 * mt + nt) + 15)
OFFA = lambda i: "v%d" % (176 + i)
OFFB = lambda i: "v%d" % (184 + i)
ADA = lambda ks: "v%d" % (168 + ks)
ADB = lambda ks: "v%d" % (172 + ks)
SLOT = 16384

# scratch SGPR roles
P, SA, SB, DA, T, T2, MA, MB, DB, CNT = ("%%[t%d]" % i for i in range(10))
#  P   ring position (slot index, even) of the k-tile being consumed
#  SA  byte base of this wave's A half-tile slot,  SB of its B half-tile slot (current k-tile)
#  DA  byte base (+ wave1k) of the ring slot that takes A0 of k-tile kt + 2 (A1 = + SLOT); DB likewise for B0 of kt + 2
#  CNT remaining steady-state trips


SPREAD = os.environ.get("P9_SPREAD", "1") == "1"   # one memory instruction per MFMA gap, counted lgkmcnt waits

# fragment read order of a k-step = order of first use by the MFMAs (mt outer, nt inner): MFMA (mt, nt) needs B[nt] and A[mt]
READ_ORDER = (("b", 0), ("a", 0), ("b", 1), ("b", 2), ("b", 3), ("a", 1), ("a", 2), ("a", 3)) if SPREAD else \
    (("b", 0), ("b", 1), ("b", 2), ("b", 3), ("a", 0), ("a", 1), ("a", 2), ("a", 3))
READ_IDX = {f: i for i, f in enumerate(READ_ORDER)}


def reads(ks, s):
    """the eight fragment reads of k-step ks into set s, in READ_ORDER"""
    return ["ds_read_b128 %s, %s offset:%d" % ((FB(s, i), ADB(ks), i * 4096) if op == "b" else (FA(s, i), ADA(ks), i * 4096))
            for op, i in READ_ORDER]


def dma(panel, i):
    """one DMA instruction: round i (0..7) of the A / B panel of the k-tile two ahead, then advance the offset by one k-tile.
    Returned as (M0 write, [load, offset advance]): one other instruction has to sit between the M0 write and the LDS-DMA
    that reads it (1 wait state)"""
    off = OFFA(i) if panel == "a" else OFFB(i)
    base = "%[pa]" if panel == "a" else "%[pb]"
    dst = DA if panel == "a" else DB
    m0 = "s_add_u32 m0, %s, %d" % (dst, (i >> 2) * SLOT + (i & 3) * 4096)
    if BUF:
        # the k-tile's byte offset rides in the scalar offset (MA / MB), advanced once per panel behind its eighth DMA
        ko = MA if panel == "a" else MB
        desc = "%[da]" if panel == "a" else "%[db]"
        rest = ["buffer_load_dwordx4 %s, %s, %s offen lds" % (off, desc, ko)]
        if i == 7:
            rest.append("s_add_u32 %s, %s, 0x80" % (ko, ko))
        return (m0, rest)
    return (m0, ["global_load_lds_dwordx4 %s, %s" % (off, base), "v_add_u32 %s, 0x80, %s" % (off, off)])


def gap(m0_sets, mids, posts):
    """instructions of one MFMA gap: M0 writes, then the reads, then the DMA that reads M0"""
    out = list(m0_sets) + list(mids)
    if m0_sets and not mids:
        out.append("s_nop 0")
    return out + list(posts)


def slot_math_top():
    """at the top of a k-tile (P = its ring position): where A0 of k-tile kt + 2 goes"""
    return ["s_add_u32 %s, %s, 8" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
            "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (T, T),
            "s_add_u32 %s, %s, %%[wave1k]" % (DA, T)]


def slot_math_advance():
    """after the barrier of k-tile kt: B0 of kt + 2 takes kt's own first slot; then P moves on to kt + 1 and the fragment
    addresses of kt + 1 are formed (their first use is the prefetch of k-step 0 inside k-step 3 of kt)"""
    out = ["s_lshl_b32 %s, %s, 14" % (T, P), "s_add_u32 %s, %s, %%[wave1k]" % (DB, T),
           "s_add_u32 %s, %s, 4" % (P, P), "s_sub_u32 %s, %s, 10" % (T2, P), "s_cmp_ge_u32 %s, 10" % P,
           "s_cselect_b32 %s, %s, %s" % (P, T2, P)]
    out += addr_math()
    return out


def addr_math():
    """SA / SB and the eight fragment address registers from P"""
    out = ["s_add_u32 %s, %s, %%[wm]" % (T, P), "s_lshl_b32 %s, %s, 14" % (SA, T),
           "s_add_u32 %s, %s, %%[wn2]" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
           "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (SB, T)]
    out += ["v_add_u32 %s, %s, %%[vl%d]" % (ADA(ks), SA, ks) for ks in range(4)]
    out += ["v_add_u32 %s, %s, %%[vl%d]" % (ADB(ks), SB, ks) for ks in range(4)]
    return out


RES_COPIES = 8        # residual variant: the first eight steady-state k-tiles each fetch four of the 32 residual quads
RES_GAPS = (3, 7, 11, 15)
RV = "v162"           # running PER-LANE byte offset of the residual row block (tile origin + lane part, advanced 16 rows per
                      # copy).  Everything sits in the VECTOR offset and the scalar offset is 0: the buffer bounds check covers
                      # voffset + the immediate only (soffset is outside it), so this is what makes rows past M read as 0 instead
                      # of reading up to 255 rows beyond R


def res_loads(c):
    """the four residual loads of copy c (row block j = c: m-tile c >> 1, 16-row half c & 1; one per n-tile), then the row
    block advances.  Quad index it = ((mt * 4 + nt) << 1) | half, as the staged epilogue numbers them."""
    mt, half = c >> 1, c & 1
    out = []
    for nt in range(4):
        it = ((mt * 4 + nt) << 1) | half
        out.append("buffer_load_dwordx4 %%[r%d], %s, %%[rs], 0 offen offset:%d" % (it, RV, nt * 64))
    return out


def ktile(dma_on, barrier_on, res_copy=None):
    """one k-tile: 64 MFMAs; the instructions of gap j go out right behind MFMA j (0..63).  res_copy = c: this k-tile also
    fetches residual row block c (four buffer loads in otherwise empty gaps; they are older than nothing the barrier of
    THIS k-tile needs, so its counted wait lets twelve instead of eight loads stay in flight)"""
    m0s = [[] for _ in range(64)]
    mids = [[] for _ in range(64)]
    posts = [[] for _ in range(64)]
    read_gaps = [[] for _ in range(4)]   # per k-step: the (relative) gaps that carry the NEXT k-step's fragment reads

    def put_dma(j, panel, i):
        if NO_DMA:
            return
        m0, rest = dma(panel, i)
        m0s[j].append(m0)
        posts[j] += rest

    pre = []
    if dma_on:
        pre += slot_math_top()
        # A(kt + 2): eight DMAs in k-steps 0 and 1 (their two slots were free all along)
        a_gaps = [int(x) for x in A_GAPS.split(",")] if A_GAPS else \
            ([4 * i + 1 for i in range(8)] if SPREAD else [4 * i + 3 for i in range(8)])
        for i in range(8):
            put_dma(a_gaps[i], "a", i)
    for ks in range(4):   # k-steps 0-2 prefetch k-steps 1-3 of the same k-tile, k-step 3 k-step 0 of the NEXT k-tile
        if ks == 3 and not barrier_on:
            continue
        # spread: one read every other gap (the DMAs take the odd gaps); k-step 2 stays dense so that every read of this
        # k-tile has returned by the barrier at the top of k-step 3 (its lgkmcnt(0) then costs nothing)
        gaps = [2 * r for r in range(8)] if (SPREAD and ks != 2) else [1 + r for r in range(8)]
        read_gaps[ks] = gaps
        for r, ins in enumerate(reads((ks + 1) & 3, (ks + 1) & 1)):
            if not NO_READS:
                mids[16 * ks + gaps[r]].append(ins)
    if barrier_on and dma_on:
        for i in range(8):   # the old k-tile's slots are free behind the barrier: B(kt + 2)
            put_dma(48 + (i if B_FAST else 1 + 2 * i), "b", i)
    if res_copy is not None:
        for g, ins in zip(RES_GAPS, res_loads(res_copy)):
            posts[g].append(ins)
        posts[RES_GAPS[-1]].append("v_add_u32 %s, %%[ldr16], %s" % (RV, RV))
    body = list(pre)
    for j in range(64):
        ks, q = j >> 4, j & 15
        mt, nt = q >> 2, q & 3
        s = ks & 1
        if q == 0 and ks == 3 and barrier_on:
            # every fragment read of this k-tile has returned; B of the next k-tile and everything older has landed (at
            # most the eight A DMAs issued above are still in flight): publish / free through ONE barrier
            body.append("s_waitcnt vmcnt(%d) lgkmcnt(0)" % ((12 if res_copy is not None else 8) if dma_on and not NO_DMA else 0))
            if not NO_BARRIER:
                body.append("s_barrier")
            body += slot_math_advance() if dma_on else advance_no_dma()
        elif not SPREAD:
            if q == 0:
                body.append("s_waitcnt lgkmcnt(0)")
        elif q in (0, 1, 2, 3, 4, 8, 12) and not (ks == 3 and barrier_on) and not NO_READS:
            # LDS returns in order: the fragment this MFMA needs is read number `need` of its k-step's eight; younger reads
            # (the rest of those eight + what this k-step has issued for the next one) may stay in flight
            need = max(READ_IDX[("b", nt)], READ_IDX[("a", mt)])
            issued_here = sum(1 for g in read_gaps[ks] if g < q)
            body.append("s_waitcnt lgkmcnt(%d)" % min(15, 7 - need + issued_here))
        body.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (ACC(mt, nt), FB(s, nt), FA(s, mt), ACC(mt, nt)))
        body += gap(m0s[j], mids[j], posts[j])
    return body


def advance_no_dma():
    out = ["s_add_u32 %s, %s, 4" % (P, P), "s_sub_u32 %s, %s, 10" % (T2, P), "s_cmp_ge_u32 %s, 10" % P,
           "s_cselect_b32 %s, %s, %s" % (P, T2, P)]
    return out + addr_math()


def setup():
    out = []
    # DMA offsets: row r = half * 128 + i * 32 + vrow (vrow = wave * 8 + lane / 8), clamped to the tile's last valid row
    for i in range(8):
        for mx, ld, off in (("%[rmaxa]", "%[lda2]", OFFA(i)), ("%[rmaxb]", "%[ldb2]", OFFB(i))):
            out += ["v_add_u32 v160, 0x%x, %%[vrow]" % (i * 32), "v_min_u32 v160, %s, v160" % mx,
                    "v_mul_lo_u32 v161, v160, %s" % ld, "v_add_u32 %s, v161, %%[vslot]" % off]
    return out


def prologue():
    out = setup()
    if BUF:
        out += ["s_mov_b32 %s, 0" % MA, "s_mov_b32 %s, 0" % MB]
    # k-tiles 0 and 1: slots 0-3 and 4-7, order A(0) B(0) A(1) B(1)
    for kt in range(2):
        out += ["s_add_u32 %s, %%[wave1k], %d" % (DA, (4 * kt) * SLOT), "s_add_u32 %s, %%[wave1k], %d" % (DB, (4 * kt + 2) * SLOT)]
        for panel in ("a", "b"):
            for i in range(8):
                m0, rest = dma(panel, i)
                out += [m0, "s_nop 0"] + rest
    # accumulators to zero while the first tiles are in flight
    out += ["v_accvgpr_write_b32 a%d, 0" % i for i in range(256)]
    out += ["s_mov_b32 %s, 0" % P] + addr_math()
    out += ["s_waitcnt vmcnt(16)", "s_barrier"]
    out += reads(0, 0)
    return out


def emit(res=False):
    global BUF
    buf_saved = BUF
    if res:
        BUF = False   # the BUF experiment (scalar k offset) only exists for the plain statement: its MA / MB are RS here
    try:
        return _emit(res)
    finally:
        BUF = buf_saved


def _emit(res):
    lines = []
    lines += prologue()
    if not res:
        lines += ["s_mov_b32 %s, %%[nloop]" % CNT, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 2f", "1:"]
        lines += ktile(True, True)
        lines += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_lg_u32 %s, 0" % CNT, "s_cbranch_scc1 1b", "2:"]
    else:
        # Residual variant (C = R + ...): the tile's residual (32 quads per lane, 128 KiB per workgroup) is fetched INSIDE the
        # K loop, four quads per k-tile over the first eight steady-state k-tiles.  Fetched in one go -- behind the loop, or in
        # front of it -- every CU asks for its 128 KiB in the same microsecond (32 MB per round of tiles) and the burst is on
        # the critical path either way (vmcnt retires in order: a counted wait covers everything older).  Short K: whatever
        # the loop did not get to is fetched by the catch-up chain (labels 1xx) in front of the last two k-tiles.
        lines += ["v_mov_b32 %s, %%[rvoff]" % RV]
        lines += ["s_mov_b32 %s, %%[nloop]" % CNT, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 100f"]
        for c in range(RES_COPIES):
            lines += ktile(True, True, res_copy=c)
            lines += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 %df" % (101 + c)]
        lines += ["1:"]
        lines += ktile(True, True)
        lines += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_lg_u32 %s, 0" % CNT, "s_cbranch_scc1 1b", "s_branch 2f"]
        for c in range(RES_COPIES):
            lines += ["%d:" % (100 + c)] + res_loads(c) + ["v_add_u32 %s, %%[ldr16], %s" % (RV, RV)]
        lines += ["%d:" % (100 + RES_COPIES), "2:"]
    lines += ktile(False, True)     # last but one: nothing left to stage, the wait drains
    lines += ktile(False, False)    # last
    lines += ["s_nop 15", "s_nop 15"]   # the last MFMAs' results before any v_accvgpr_read of the epilogue
    return lines


def main():
    lines = emit()
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("P9_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "gemm_p9_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_p9.py -- do not edit.  The main loop of GEMM schedule 9 as one asm statement.\n")
        for name, ls in (("ALG_GEMM_P9_LOOP_ASM", lines), ("ALG_GEMM_P9_LOOP_ASM_RES", emit(res=True))):
            f.write("#define %s \\\n" % name)
            for ln in ls:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
        regs = ["a%d" % i for i in range(256)] + ["v%d" % i for i in range(160, 256)]
        f.write("#define ALG_GEMM_P9_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
        f.write("#define ALG_GEMM_P9_ACC_CLOBBERS \\\n  " + ", ".join('"a%d"' % i for i in range(256)) + '\n')
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    print("wrote", os.path.normpath(path), len(lines), "lines,", n_mfma, "MFMAs")


if __name__ == "__main__":
    main()
