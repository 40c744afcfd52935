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

# experiment knobs (scripts/experiments/p9_build_variant.sh): timing ablations produce garbage results
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


# ---------------------------------------------------------------------------------------------------------------------
# B-PACKED form (round 5): the B operand (the weight of a linear layer) never touches the LDS.
#
# Ablation (docs/lab_notebook_r5.md item 20): the loop's two LDS streams -- LDS-DMA in, fragment reads out -- cost 17 % and 18 % of it
# and nothing else does.  A weight is static, so it can be stored ONCE in the order the MFMA wants it: fragment (n-block of 32
# rows, k-step of 16) = 1 KiB, lane l holds row l & 31, elements 8 (l >> 5) .. + 8 of the k-step (alg_pack_b_bf16: chunk index =
# n-block * K / 16 + k-step).  A fragment is then ONE coalesced buffer_load_dwordx4 straight into the registers the MFMA reads:
# no DMA, no LDS slot, no ds_read for B -- half of both streams gone.
#   registers  BK(p, ks, nt) = v[32 + 64 p + 16 ks + 4 nt .. + 3], p = k-tile parity: TWO k-tiles of B (128 registers; the plain
#              statement's FB sets are unused).  The load of (k-tile kt + 2, ks, nt) goes out right behind the last MFMA of k-tile
#              kt that reads BK(kt & 1, ks, nt) (mt = 3: gap 16 ks + 12 + nt): two k-tiles (128 MFMAs, > 2 us) ahead of its use.
#              The vector-memory counter retires IN ORDER and counts the A panel's LDS-DMA too: with one k-tile of B prefetch every
#              counted wait for B also forces the A DMAs issued just before it to land within a k-tile; with two, everything has two.
#   addresses  v[184 + nt] = lane * 16 + chunk base of the wave's n-block nt (operand bvo + nt * operand bnt); the k-step is the
#              instruction offset (ks * 1024), the k-tile the scalar offset (MB = 4096 (kt + 2)); descriptor operand db
#              (num_records = bytes of the packed weight: n-blocks past N read as zeros)
#   waits      one counted vmcnt in front of every k-step, its immediate = the vector-memory instructions issued since the youngest
#              one it needs (BK(., ks, 3); at the barrier also the last DMA of A(kt + 1)): computed by simulation of the issue order
#              over every context a k-tile can run in (bpk_waits()).
#   A panel    LDS-DMA THREE k-tiles ahead: with B out of the LDS the ten slots hold A alone -- k-tile kt sits in slots 4 kt mod 10
#              (+1) and the pair 4 kt + 2 (+3), the B slots of the plain form, takes A(kt + 3); fragment reads one k-step ahead, one
#              barrier per k-tile, as in the plain form.
#   loop       the body is a PAIR of k-tiles (parity 0, 1); (K / 64 - 4) / 2 trips, then the last four or five k-tiles (K / 64 even /
#              odd) as straight code.  K >= 256.
BK = lambda p, ks, nt: "v[%d:%d]" % (32 + 64 * p + 16 * ks + 4 * nt, 32 + 64 * p + 16 * ks + 4 * nt + 3)
VBN = lambda nt: "v%d" % (184 + nt)
BPK_A_GAPS = (17, 20, 23, 26, 33, 36, 39, 42)     # the eight A DMAs of a k-tile (no gap shared with a B load: those sit in 12..15 + 16 ks)


def b_load(p, ks, nt):
    return "buffer_load_dwordx4 %s, %s, %%[db], %s offen offset:%d" % (BK(p, ks, nt), VBN(nt), MB, ks * 1024)


def bpk_addr_math():
    """SA and the four A fragment address registers from P"""
    out = ["s_add_u32 %s, %s, %%[wm]" % (T, P), "s_lshl_b32 %s, %s, 14" % (SA, T)]
    return out + ["v_add_u32 %s, %s, %%[vl%d]" % (ADA(ks), SA, ks) for ks in range(4)]


def bpk_advance():
    return ["s_add_u32 %s, %s, 4" % (P, P), "s_sub_u32 %s, %s, 10" % (T2, P), "s_cmp_ge_u32 %s, 10" % P,
            "s_cselect_b32 %s, %s, %s" % (P, T2, P)] + bpk_addr_math()


def ktile_bpk(p, dma_on, barrier_on, b_next, waits):
    """one k-tile of the B-packed form at parity p.  dma_on: stage A(kt + 3); barrier_on: a next k-tile exists (publish / free the
    ring); b_next: fetch B(kt + 2); waits[ks]: the vmcnt immediate in front of k-step ks"""
    m0s = [[] for _ in range(64)]
    mids = [[] for _ in range(64)]
    posts = [[] for _ in range(64)]
    pre = []
    if dma_on:
        # where A0 of k-tile kt + 3 goes: ring position (P + 12) mod 10 = (P + 2) mod 10
        pre += ["s_add_u32 %s, %s, 2" % (T, P), "s_sub_u32 %s, %s, 10" % (T2, T), "s_cmp_ge_u32 %s, 10" % T,
                "s_cselect_b32 %s, %s, %s" % (T, T2, T), "s_lshl_b32 %s, %s, 14" % (T, T), "s_add_u32 %s, %s, %%[wave1k]" % (DA, T)]
        for i in range(8):
            m0, rest = dma("a", i)
            m0s[BPK_A_GAPS[i]].append(m0)
            posts[BPK_A_GAPS[i]] += rest
    if b_next:
        for ks in range(4):
            for nt in range(4):
                posts[16 * ks + 12 + nt].append(b_load(p, ks, nt))
    posts[63].append("s_add_u32 %s, %s, 0x1000" % (MB, MB))
    for ks in range(4):    # A fragments of the NEXT k-step (k-step 3: of the next k-tile, behind the barrier) in gaps 0, 2, 5, 8
        if ks == 3 and not barrier_on:
            continue
        for r, g in enumerate((0, 2, 5, 8)):
            mids[16 * ks + g].append("ds_read_b128 %s, %s offset:%d" % (FA((ks + 1) & 1, r), ADA((ks + 1) & 3), r * 4096))
    body = list(pre)
    for j in range(64):
        ks, q = j >> 4, j & 15
        mt, nt = q >> 2, q & 3
        if q == 0:
            # this k-step's A fragments (read during the previous k-step) and B fragments (loaded two k-tiles ago) are in registers
            body.append("s_waitcnt vmcnt(%d) lgkmcnt(0)" % waits[ks])
            if ks == 3 and barrier_on:
                body.append("s_barrier")
                body += bpk_advance()
        body.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (ACC(mt, nt), BK(p, ks, nt), FA(ks & 1, mt), ACC(mt, nt)))
        body += gap(m0s[j], mids[j], posts[j])
    return body


def bpk_prologue():
    L = []
    for i in range(8):
        L += ["v_add_u32 v160, 0x%x, %%[vrow]" % (i * 32), "v_min_u32 v160, %[rmaxa], v160",
              "v_mul_lo_u32 v161, v160, %[lda2]", "v_add_u32 %s, v161, %%[vslot]" % OFFA(i)]
    L += ["v_mov_b32 %s, %%[bvo]" % VBN(0)] + ["v_add_u32 %s, %%[bnt], %s" % (VBN(nt), VBN(nt - 1)) for nt in range(1, 4)]
    L += ["s_mov_b32 %s, 0" % MB]
    for kt, slot in ((0, 0), (1, 4), (2, 8)):
        L += ["s_add_u32 %s, %%[wave1k], %d" % (DA, slot * SLOT)]
        for i in range(8):
            m0, rest = dma("a", i)
            L += [m0, "s_nop 0"] + rest
        if kt == 0:      # B(0), B(1) right behind A(0): the final wait lets A(1) and A(2) fly
            for pk in range(2):
                L += [b_load(pk, ks, nt) for ks in range(4) for nt in range(4)] + ["s_add_u32 %s, %s, 0x1000" % (MB, MB)]
    L += ["v_accvgpr_write_b32 a%d, 0" % i for i in range(256)]
    L += ["s_mov_b32 %s, 0" % P] + bpk_addr_math()
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    L += ["ds_read_b128 %s, %s offset:%d" % (FA(0, r), ADA(0), r * 4096) for r in range(4)]
    return L


def bpk_events(lines, kt, prologue=False):
    """the vector-memory instructions of `lines` in issue order, tagged with what they fetch: ('b', k-tile, ks, nt) / ('a', k-tile, i);
    counted waits as ('wait', ks).  kt: the k-tile the lines compute (B loads fetch kt + 2, A DMAs kt + 3); the prologue fetches
    A(0), B(0), B(1), A(1), A(2) in that order"""
    ev, na, nb = [], 0, 0
    for ln in lines:
        if ln.startswith("buffer_load_dwordx4 v[") and not ln.endswith("lds"):
            reg = int(ln.split("[")[1].split(":")[0]) - 32
            tgt = (nb // 16) if prologue else kt + 2
            ev.append(("b", tgt, (reg % 64) // 16, (reg % 16) // 4))
            nb += 1
        elif ln.startswith("global_load_lds"):
            ev.append(("a", (na // 8) if prologue else kt + 3, na % 8))
            na += 1
        elif ln.startswith("s_waitcnt") and "vmcnt" in ln and not prologue:
            ev.append(("wait", sum(1 for e in ev if e[0] == "wait")))
    return ev


def bpk_waits(kinds, first_kt_options):
    """vmcnt immediates for a straight sequence of k-tiles `kinds` = [(p, dma_on, barrier_on, b_next)]: for every context the sequence
    can run in (first_kt_options: 0 = right behind the prologue, 2 = behind a steady-state pair) the issue order is simulated and
    the count of vector-memory instructions issued after the youngest REQUIRED one is taken; the minimum over the contexts is safe
    in all of them."""
    result = [[63] * 4 for _ in kinds]
    for first in first_kt_options:
        hist = []
        if first == 0:
            hist += bpk_events(bpk_prologue(), 0, prologue=True)
        else:      # two steady-state k-tiles in front (their own predecessors do not matter: everything older is older still)
            for i in range(2):
                hist += [e for e in bpk_events(ktile_bpk(i & 1, True, True, True, [63] * 4), first - 2 + i) if e[0] != "wait"]
        for n, kind in enumerate(kinds):
            kt = first + n
            for e in bpk_events(ktile_bpk(*kind, waits=[63] * 4), kt):
                if e[0] != "wait":
                    hist.append(e)
                    continue
                ks = e[1]
                need = [("b", kt, ks, 3)] + ([("a", kt + 1, 7)] if (ks == 3 and kind[2]) else [])
                pos = max(hist.index(x) for x in need)          # (ValueError = the schedule asks for something never fetched)
                result[n][ks] = min(result[n][ks], len(hist) - 1 - pos)
    return result


def emit_bpk():
    S0, S1 = (0, True, True, True), (1, True, True, True)
    L = bpk_prologue()
    # ---- (K / 64 - 4) / 2 steady-state pairs (operand nloop = K / 64 - 2 >= 2) ----
    L += ["s_sub_u32 %s, %%[nloop], 2" % CNT, "s_lshr_b32 %s, %s, 1" % (CNT, CNT), "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc1 2f", "1:"]
    w = bpk_waits([S0, S1], (0, 2))
    L += ktile_bpk(*S0, waits=w[0]) + ktile_bpk(*S1, waits=w[1])
    L += ["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_lg_u32 %s, 0" % CNT, "s_cbranch_scc1 1b", "2:"]
    # ---- the last four (K / 64 even) or five (odd) k-tiles: A(kt + 3) / B(kt + 2) only while they exist ----
    L += ["s_and_b32 %s, %%[nloop], 1" % CNT, "s_cmp_eq_u32 %s, 0" % CNT, "s_cbranch_scc0 5f"]
    for label, r in ((None, 4), ("5:", 5)):
        if label:
            L += [label]
        kinds = [(i & 1, i + 3 < r, i + 1 < r, i + 2 < r) for i in range(r)]
        w = bpk_waits(kinds, (0, 2))
        for kind, wk in zip(kinds, w):
            L += ktile_bpk(*kind, waits=wk)
        if r == 4:
            L += ["s_branch 6f"]
    L += ["6:", "s_nop 15", "s_nop 15"]
    return L


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
        for name, ls in (("ALG_GEMM_P9_LOOP_ASM", lines), ("ALG_GEMM_P9_LOOP_ASM_RES", emit(res=True)), ("ALG_GEMM_P9_LOOP_ASM_BPK", emit_bpk())):
            f.write("#define %s \\\n" % name)
            for ln in ls:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
        regs = ["a%d" % i for i in range(256)] + ["v%d" % i for i in range(160, 256)]
        f.write("#define ALG_GEMM_P9_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
        regs_bpk = ["a%d" % i for i in range(256)] + ["v%d" % i for i in range(32, 256)]
        f.write("#define ALG_GEMM_P9_BPK_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs_bpk) + '\n')
        f.write("#define ALG_GEMM_P9_ACC_CLOBBERS \\\n  " + ", ".join('"a%d"' % i for i in range(256)) + '\n')
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    print("wrote", os.path.normpath(path), len(lines), "lines,", n_mfma, "MFMAs")


if __name__ == "__main__":
    main()
