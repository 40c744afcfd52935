#!/usr/bin/env python3
"""Generates alg_amd/csrc/attn128_q64_loop.inc (Cfg(64) -- the same construction at half the head width -- is kept as a generator mode): the
steady-state KV loop of the 64-QUERIES-PER-WAVE flash attention as ONE inline-asm statement.

Why (profiles/r4_pmc_d128_q64_vs_pipe.txt): with 64 queries per wave every K / V^T fragment read from LDS feeds
TWO MFMAs (the wave's two 32-query halves) -- half the LDS instructions and half the L2 -> LDS traffic per FLOP of the 32-query
statements (gen_attn_pipe.py, gen_attn128_pipe.py) -- but round 4's 64-query kernel left the order of everything between its asm
MFMAs to hipcc (matrix pipe busy 62 % of its cycles, and a three-round hunt for a register the compiler recycled in the shadow
of an MFMA it cannot see).  Here every instruction of the loop has a fixed place between two MFMAs, as in the 32-query
statements, and no compiler-scheduled instruction exists between the first and the last MFMA of the statement.

One wave = 64 queries (two halves qh = 0, 1 of 32), a workgroup = 4 waves = 256 queries, one wave per SIMD.  Software pipeline
over KV tiles of 64 keys; iteration t issues, per wave (d = 128 | d = 64):
    PV(t-1):  O^T[qh][dt] += V(t-1)^T[dt][kk] P(t-1)^T[qh][kk]      32 | 16 MFMAs, 16 | 8 V^T fragment reads, C/D = the O operands
    QK(t+1):  S(t+1)^T[sub][qh] = K(t+1)[sub] Q[qh]^T                32 | 16 MFMAs, 16 | 8 K fragment reads, first k-step from C = 0
    softmax(t): 32 score pairs per lane (64 keys x 64 queries / 64 lanes / 2), spread one pair per two | one MFMA gaps:
                f = s * c - m * c (v_fma; dropped in the pre-scaled zero-offset form), e = exp2(f), P = cvt_pk(e, e'), row sum
The running max m is the LAZY one of the frame (attention128_q64.hip): probabilities are formed against the current m, the
statement leaves (code 1) as soon as a tile's row sum leaves [0, 2^80), the exact path (tile max, rescale) stays in C++.

Collective protocol (identical in the C++ loop around the statement, so the waves of a workgroup may be inside or outside
independently): top of iteration t   s_waitcnt vmcnt(NP); s_barrier; DMA K(t+3) -> K slot (t+3) & 3, V^T(t+2) -> V slot (t+2) & 3
(NP = 8 | 4 pieces of 1 KiB per wave and iteration; the counted wait leaves the previous iteration's pieces in flight).
Unrolled x 4 (ring slots as immediates).  Entered at t = 1 (mod 4) through a warm-up (QK(t) alone, then QK(t+1) under
softmax(t)); runs iterations t < tend (tend = the last tile QK may compute unmasked; at least iteration t of the entry must be
allowed) and leaves through the drain of whatever phase would come next: PV of its last tile.  See emit() for the operand list.

Register plan (literal names, clobbered):
    v[VB : VB+63] SA, v[VB+64 : VB+127] SB    score tiles [sub][qh][16]; roles alternate with t & 1
    v[VB+128 : +159] PA, v[VB+160 : +191] PB  packed probabilities [qh][kk][4]
    v[VB+192 ..] TS0 TS1 EA0 EA1 EB0 EB1 F0 F1
    a[QA ..] Q fragments [qh][ks][4], a[FR : FR+31] eight fragment buffers (ring); O^T: 16-register "+a" operands o0 .. o(2 DT - 1)
"""
import os
import sys


class Cfg:
    def __init__(self, d, fma=True, name=None):
        self.d = d
        self.KS = d // 16            # k-steps of QK
        self.DT = d // 32            # 32-row d-tiles of O^T
        self.TILE = 64 * d * 2       # bytes of a K tile = bytes of a V^T tile
        self.KSUB = 32 * d * 2       # bytes of a 32-key sub-tile of K
        self.NP = 2 * self.TILE // 1024 // 4     # DMA pieces per wave and iteration (K + V^T)
        self.fma = fma               # False: scores arrive pre-scaled in log2 units with a zero offset (d = 64 frame)
        self.VB = 52                 # first literal ArchVGPR
        self.SA, self.SB = self.VB, self.VB + 64
        self.PA, self.PB = self.VB + 128, self.VB + 160
        t = self.VB + 192
        self.TS = (t, t + 1)
        self.EA = (t + 2, t + 3)
        self.EB = (t + 4, t + 5)
        self.F0, self.F1 = t + 6, t + 7
        self.VEND = t + 8
        self.NO = 2 * self.DT        # O operands (16 registers each)
        self.QA = 128                # AccVGPR layout: O operands are the compiler's (outside [QA, AEND)), Q, fragment ring
        self.FR = self.QA + 2 * self.KS * 4
        self.AEND = self.FR + 32
        self.name = name or ("ATTN%d_Q64" % d)
        self.n_frag = 2 * self.KS + 4 * self.DT      # fragments per full iteration (K: sub x ks, V: kk x dt)
        self.n_mfma = 2 * self.n_frag
        assert self.VEND <= 256 and self.AEND <= 256


v = lambda i: "v%d" % i
vr = lambda i, n: "v[%d:%d]" % (i, i + n - 1)
ar = lambda i, n: "a[%d:%d]" % (i, i + n - 1)
KD, VD = 88, 92       # literal SGPR quads: the raw buffer descriptors of the K / V^T tile being fetched (clobbered)
AHEAD = int(os.environ.get("ATTN_Q64_AHEAD", "4"))     # fragment reads in flight ahead of the fragment being multiplied (ring: 8 buffers)
# timing-only experiment knobs (WRONG RESULTS; never set for the committed .inc): "noadd" drops the row-sum adds, "ones" issues the
# MFMAs a ones-row of V^T would cost, "nofma" (d = 128) generates the pre-scaled zero-offset form behind the unchanged frame
EXP = set(filter(None, os.environ.get("ATTN_Q64_EXP", "").split(",")))
DMA_STRIDE = int(os.environ.get("ATTN_Q64_DMA_STRIDE", "2"))    # a DMA piece every DMA_STRIDE-th gap (M0 write, then the load one gap later)
DMA_SHIFT = int(os.environ.get("ATTN_Q64_DMA_SHIFT", "0"))      # first gap that carries a DMA half


def frag_read(c, buf, which, slot, half, step):
    """ds_read_b128 of one fragment into ring buffer `buf`: K sub-tile `half` k-step `step`, or V^T d-tile `half` kv block `step`.
    The lane part (operand lk<step> / lv<step>) carries the ring base and the swizzled 16-byte slot of the step."""
    off = slot * c.TILE + half * (c.KSUB if which == "k" else 4096)
    return "ds_read_b128 %s, %%[l%s%d] offset:%d" % (ar(c.FR + 4 * buf, 4), which, step, off)


def pair_regs(c, S, n):
    """score pair n (0..31) of the tile in S: (register of the first score, qh, P register index within the 32-register P tile)
    order: qh fastest (the two row sums alternate), then j (pair inside a register quad), g (quad), sub"""
    qh, r = n & 1, n >> 1
    j, g, sub = r & 3, (r >> 2) & 1, r >> 3
    s0 = S + (sub * 2 + qh) * 16 + 8 * g + 2 * j
    kk = sub * 2 + g
    return s0, qh, qh * 16 + kk * 4 + j


def softmax_stream(c, S, P):
    """The VALU work of softmax(t) as `pre` (in front of the barrier), 64 half-pair groups and `post` (behind the last MFMA):
        group 2n:     exp ea(n) | fma fb(n) | cvt P(n-1) | row sum += eb(n-1)
        group 2n + 1: exp eb(n) | fma fa(n+1) | row sum += ea(n)
    one transcendental per group; every result is read at least one group (= one MFMA in the steady state) after it was written."""
    groups = []
    pend = []
    first_sum = {0: True, 1: True}

    def add(qh, reg):
        if first_sum[qh]:
            first_sum[qh] = False
            return "v_mov_b32 %s, %s" % (v(c.TS[qh]), v(reg))
        if "noadd" in EXP or "mfma4" in EXP:
            return None
        return "v_add_f32 %s, %s, %s" % (v(c.TS[qh]), v(c.TS[qh]), v(reg))

    def fma(dst, src, qh):
        return "v_fma_f32 %s, %s, %%[c], %%[negmc%d]" % (v(dst), v(src), qh)

    s0, qh0, _ = pair_regs(c, S, 0)
    pre = [fma(c.F0, s0, qh0)] if c.fma else []
    for n in range(32):
        s0, qh, _ = pair_regs(c, S, n)
        ea, eb = c.EA[n & 1], c.EB[n & 1]
        g = ["v_exp_f32 %s, %s" % (v(ea), v(c.F0 if c.fma else s0))]
        if c.fma:
            g.append(fma(c.F1, s0 + 1, qh))
        if n >= 1:
            _, qp, pp = pair_regs(c, S, n - 1)
            g.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P + pp), v(c.EA[(n - 1) & 1]), v(c.EB[(n - 1) & 1])))
            g.append(add(qp, c.EB[(n - 1) & 1]))
            if "mfma4" in EXP and (pp & 1) == 1:
                # EXPERIMENT (timing only so far): row sums on the matrix pipe -- v_mfma_f32_4x4x4_16b_bf16 with A = ones sums the four
                # probabilities of two packed P registers per lane (2 passes); issued one group behind the cvt that completes the pair
                pend.append("v_mfma_f32_4x4x4_16b_bf16 %s, %s, %s, %s" % (ar(c.AEND + 4 * qp, 4), ar(c.AEND + 8, 2), vr(P + pp - 1, 2), ar(c.AEND + 4 * qp, 4)))
        groups.append(g)
        g = ["v_exp_f32 %s, %s" % (v(eb), v(c.F1 if c.fma else s0 + 1))]
        if c.fma and n + 1 < 32:
            s1, q1, _ = pair_regs(c, S, n + 1)
            g.append(fma(c.F0, s1, q1))
        g.append(add(qh, ea))
        g += pend
        del pend[:]
        groups.append(g)
    _, qp, pp = pair_regs(c, S, 31)
    post = ["s_nop 1", "v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P + pp), v(c.EA[1]), v(c.EB[1])), add(qp, c.EB[1])]
    groups = [[x for x in g if x is not None] for g in groups]
    return pre, groups, [x for x in post if x is not None]


def top_protocol(c, phase):
    """(head, DMA pieces) of iteration t, t & 3 == phase: all but the previous iteration's NP pieces have landed (K(t+1), V(t) and
    older); behind the barrier this wave's pieces of K(t+3) and V^T(t+2).  A piece is (M0 write, [load], scalar advance or None):
    the two halves go into two consecutive MFMA gaps, so the M0 write is separated from the LDS-DMA that reads it by real work
    instead of an s_nop; the advance of the tile offset goes into the gap behind the last piece of each panel."""
    ks, vs = (phase + 3) & 3, (phase + 2) & 3
    head = ["s_waitcnt vmcnt(%d)" % c.NP, "s_barrier"]
    if "nobarrier" in EXP:
        head = head[:1]
    if "nowait" in EXP:
        head = []
    pieces = []
    # A piece is a buffer-addressed LDS-DMA: per-lane offset (kvo / vvo: the lane's place inside a tile, constant) against a raw
    # buffer descriptor whose BASE is the tile (KD = s[88:91] for K(t+3), VD = s[92:95] for V^T(t+2), literal SGPRs loaded from the
    # kd0..3 / vd0..3 operands at entry) and whose num_records is what is left of the (batch, head) panel from there.  The tile
    # advance is four scalar instructions behind the last piece of each panel (base += step with carry, num_records -= step,
    # saturating at 0), so a piece whose tile lies past the end of the panel -- the statement prefetches three / two tiles ahead of
    # the last tile it computes -- fetches nothing and stores zeros that nobody reads, and K rows past the end inside the ragged
    # last tile arrive as zeros (the frame, which keeps that tile, masks them).  Only the per-lane offset is range-checked by the
    # hardware, which is why the tile lives in the base and not in the instruction's scalar offset.
    # (Round 5, first form: global_load_lds with a v_add and a v_min clamp per piece -- 16 VALU instructions per iteration that the
    # scalar form does not issue.)
    half = c.NP // 2

    def advance(d0, step):
        return ["s_add_u32 s%d, s%d, %s" % (d0, d0, step), "s_addc_u32 s%d, s%d, 0" % (d0 + 1, d0 + 1),
                "s_sub_u32 s%d, s%d, %s" % (d0 + 2, d0 + 2, step), "s_cselect_b32 s%d, 0, s%d" % (d0 + 2, d0 + 2)]
    for r in range(half):
        pieces.append(("s_add_u32 m0, %%[wk], %d" % (ks * c.TILE + r * 4096),
                       ["buffer_load_dwordx4 %%[kvo%d], s[%d:%d], 0 offen lds" % (r, KD, KD + 3)],
                       advance(KD, "%[kstep]") if r == half - 1 else []))
    for r in range(half):
        pieces.append(("s_add_u32 m0, %%[wv], %d" % (vs * c.TILE + r * 4096),
                       ["buffer_load_dwordx4 %%[vvo%d], s[%d:%d], 0 offen lds" % (r, VD, VD + 3)],
                       advance(VD, "0x80") if r == half - 1 else []))
    if "nodma" in EXP:
        pieces = []
    return head, pieces


def flat_pieces(pieces):
    out = []
    for m0, rest, advance in pieces:
        out += [m0, "s_nop 0"] + rest + advance
    return out


def frag_order(c, pv, qk):
    """fragment stream of one iteration: the first AHEAD V^T fragments (one | two kv blocks: barrier-free, prefetched by the
    previous iteration), then K and the middle kv blocks interleaved (K, K, V), the last kv block at the end -- S(t+1) is
    complete a kv block's worth of MFMAs before the next iteration's softmax reads it, and an accumulator is touched again at the
    earliest four MFMAs later."""
    ks_frags = [("k", sub, ks) for ks in range(c.KS) for sub in range(2)] if qk else []
    vfr = lambda kk: [("v", dt, kk) for dt in range(c.DT)]
    if not pv:
        return ks_frags
    if not qk:
        return [f for kk in range(4) for f in vfr(kk)]
    lead = -(-AHEAD // c.DT)                      # kv blocks that make up the first AHEAD fragments
    out = [f for kk in range(lead) for f in vfr(kk)]
    mid = [f for kk in range(lead, 3) for f in vfr(kk)]
    i = 0
    while i < len(ks_frags) or mid:
        out += ks_frags[i:i + 2]
        i += 2
        if mid:
            out.append(mid.pop(0))
    return out + vfr(3)


def first_reads(c, phase):
    """the first AHEAD fragment reads of a full iteration at `phase`: V^T(t-1), kv block 0 (| 0 and 1) -- independent of the
    iteration's barrier (the tile landed two iterations earlier), so the PREVIOUS iteration issues them behind its last fragments"""
    vslot = (phase - 1) & 3
    fr = frag_order(c, True, True)[:AHEAD]
    assert all(kind == "v" for kind, _, _ in fr)
    return [frag_read(c, j, "v", vslot, dt, kk) for j, (_, dt, kk) in enumerate(fr)]


def iteration(c, phase, X, Y, U, W, pv=True, softmax=True, qk=True, reads_in_flight=0, prefetch_next=None, dma=None):
    """one iteration at ring phase t & 3 == phase.  X: S(t) (read by the softmax), Y: S(t+1) (written by QK), U: P(t-1) (PV's B
    operand), W: P(t) (written by the softmax).  reads_in_flight: how many of its first fragment reads the caller has issued.
    MFMA gap g (the instructions behind MFMA g): the fragment read that reuses a ring buffer (behind the second MFMA of a
    fragment), half a DMA piece (even g: M0, odd g: the LDS-DMA and its offset advance), its share of the softmax's half-pair
    groups (spread evenly over the gaps).  Fragment waits: one counted lgkmcnt per TWO fragments (reads run four fragments ahead:
    lgkmcnt(2) in front of fragment j leaves j + 2, j + 3 in flight)."""
    kslot, vslot = (phase + 1) & 3, (phase - 1) & 3
    fr = frag_order(c, pv, qk)
    n_f = len(fr)
    assert n_f % 2 == 0

    def read(j):
        kind, half, step = fr[j]
        return frag_read(c, j % 8, kind, kslot if kind == "k" else vslot, half, step)

    head = [read(j) for j in range(reads_in_flight, min(AHEAD, n_f))]
    nxt = first_reads(c, prefetch_next) if prefetch_next is not None else None
    # ---- the MFMA stream: (instructions in front, the MFMA, the read behind it) ----
    stream, seen = [], set()
    for j, (kind, half, step) in enumerate(fr):
        pre = []
        if j % 2 == 0:
            issued = (n_f + AHEAD if nxt is not None else n_f) - 1           # index of the last read that will ever be issued
            younger = min(j + AHEAD - 1, issued) - (j + 1)                   # reads issued behind read j + 1 at this point
            if "nolgkm" not in EXP:
                pre.append("s_waitcnt lgkmcnt(%d)" % max(younger, 0))
        frag = ar(c.FR + 4 * (j % 8), 4)
        for qh in range(2):
            if kind == "k":
                acc = vr(Y + (half * 2 + qh) * 16, 16)
                cin = acc if (half, qh) in seen else "0"
                seen.add((half, qh))
                text = "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, frag, ar(c.QA + (qh * c.KS + step) * 4, 4), cin)
            else:
                acc = "%%[o%d]" % (qh * c.DT + half)
                text = "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, frag, vr(U + qh * 16 + step * 4, 4), acc)
            behind = []
            if qh == 1:      # the fragment's buffer is free again four fragments from now: the read that reuses it
                if j + AHEAD < n_f:
                    behind.append(read(j + AHEAD))
                elif nxt is not None:
                    behind.append(nxt[j + AHEAD - n_f])
            stream.append((pre if qh == 0 else [], text, behind))
            if "ones" in EXP and kind == "v" and qh == 1 and half == c.DT - 1:
                # EXPERIMENT (timing only): the row sums on the matrix pipe -- one more MFMA per (kv block, query half) with P as
                # the B operand, as a ones-row of V^T would cost
                for q2 in range(2):
                    acc = ar(c.AEND + 16 * q2, 16)
                    stream.append(([], "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, ar(c.QA, 4), vr(U + q2 * 16 + step * 4, 4), acc), []))
    # ---- the softmax's groups, spread evenly over the gaps ----
    groups, post = [], []
    if softmax:
        _, groups, post = softmax_stream(c, X, W)
    n_m = len(stream)
    lines = list(head)
    gi = 0
    for g, (pre, text, behind) in enumerate(stream):
        lines += pre
        lines.append(text)
        lines += behind
        gg = g - DMA_SHIFT
        piece = dma[gg // DMA_STRIDE] if dma and gg >= 0 and gg % DMA_STRIDE < 2 and gg // DMA_STRIDE < len(dma) else None
        # the scalar tile advance behind the last K / V^T piece of the iteration: base += step (add + carry) in the gap behind the
        # load, num_records -= step (subtract + saturate) in the one after; each pair reads the SCC its first instruction wrote
        for back, part in ((2, slice(0, 2)), (3, slice(2, 4))):
            q, rem = divmod(gg - back, DMA_STRIDE)
            if dma and gg >= back and rem == 0 and q < len(dma):
                lines += dma[q][2][part]
        if piece and gg % DMA_STRIDE == 0:
            lines.append(piece[0])
        upto = (g + 1) * len(groups) // n_m
        while gi < upto:
            lines += groups[gi]
            gi += 1
        if piece and gg % DMA_STRIDE == 1:
            lines += piece[1]
    assert gi == len(groups)
    return lines + post


def check_and_count(c, fail_label):
    """row-sum check of the iteration just issued: both query halves at once through their sum (inf and NaN propagate; two sums
    below 2^79 each would be the exact condition, 2^80 for their total is the same guard one bit earlier), then l += tile sums,
    t += 1"""
    out = ["v_add_f32 %s, %s, %s" % (v(c.F1), v(c.TS[0]), v(c.TS[1])), "v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(c.F1),   # !(2^80 > sum)
           "s_nop 4", "s_cbranch_vccnz %s" % fail_label]
    out += ["v_add_f32 %%[l%d], %%[l%d], %s" % (qh, qh, v(c.TS[qh])) for qh in range(2)]
    return out + ["s_add_u32 %[t], %[t], 1"]


def emit(c):
    """Operands of the statement:
        o0 .. o(2 DT - 1) "+a" f32x16   O^T accumulators [qh][dt]
        l0, l1 "+v"                      running row sums of the two query halves (partial per half-wave lane)
        t "+s"                           in: the first iteration (= 1 mod 4); out: the iteration the frame continues with
        code "=&s"                       0: t's top-of-iteration protocol NOT done; 1: row-sum check failed in iteration t (its protocol,
                                         PV(t-1) and QK(t+1) are done, softmax(t) is not: the frame redoes tile t from QK)
        lk0 .. lk(KS-1), lv0 .. lv3 "v"  LDS byte address of the lane's K / V^T fragment per k-step / kv block (ring base included)
        kvo0 .., vvo0 .. "v"             the lane's byte offset inside a K / V^T tile, one per DMA piece
        kd0..3, vd0..3 "s"               raw buffer descriptors at entry: base = K tile t + 3 / V^T tile t + 2 of the (batch, head) panel,
                                         num_records = bytes of the panel from there (0 when the tile lies past the end), word 3 = 0x20000
        qvo0, qvo1 "v", qb "s" (64 bit)  Q rows of the two query halves
        negmc0, negmc1 "v", c "s"        (fma form) -m c per query half and the score scale in log2 units
        kstep "s" bytes per K tile, tend "s" (iterations t < tend run), wk / wv "s" LDS address of the rings + wave * 1024"""
    L = []
    # ---- entry: Q fragments of both query halves ----
    for qh in range(2):
        L += ["global_load_dwordx4 %s, %%[qvo%d], %%[qb] offset:%d" % (ar(c.QA + (qh * c.KS + ks) * 4, 4), qh, 32 * ks)
              for ks in range(c.KS)]
    L += ["s_mov_b32 s%d, %%[kd%d]" % (KD + i, i) for i in range(4)] + ["s_mov_b32 s%d, %%[vd%d]" % (VD + i, i) for i in range(4)]
    roles = {1: (c.SA, c.SB, c.PA, c.PB), 2: (c.SB, c.SA, c.PB, c.PA), 3: (c.SA, c.SB, c.PA, c.PB), 0: (c.SB, c.SA, c.PB, c.PA)}
    if "mfma4" in EXP:
        L += ["v_mov_b32 %s, 0x3f803f80" % v(c.F0)] + ["v_accvgpr_write_b32 a%d, %s" % (c.AEND + 8 + i, v(c.F0)) for i in range(2)]
        L += ["v_accvgpr_write_b32 a%d, 0" % (c.AEND + i) for i in range(8)]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]   # Q (and, once, whatever the caller had in flight)
    # ---- warm-up at phase 1: iteration t's protocol, QK(t) alone into X, then QK(t+1) under softmax(t) ----
    head, pieces = top_protocol(c, 1)
    L += head + flat_pieces(pieces)
    X, Y, U, W = roles[1]
    L += iteration(c, 0, Y, X, U, W, pv=False, softmax=False)       # phase 0's "next" K slot is slot 1 = t & 3: S(t) -> X
    L += ["s_nop 15", "s_nop 15"]                                     # S(t) complete before the first VALU reads it
    pre, _, _ = softmax_stream(c, X, W)
    L += pre
    nxt = first_reads(c, 2)
    L += iteration(c, 1, X, Y, U, W, pv=False, prefetch_next=2 if nxt else None)
    L += check_and_count(c, "90f")
    L += ["s_cmp_ge_u32 %[t], %[tend]", "s_cbranch_scc1 22f", "s_branch 12f"]
    # ---- the loop: phases 1, 2, 3, 0; behind every iteration: leave through the drain of the NEXT phase when t has reached tend ----
    L += ["11:"]
    for ph in (1, 2, 3, 0):
        if ph == 2:
            L += ["12:"]
        X, Y, U, W = roles[ph]
        head, pieces = top_protocol(c, ph)
        pre, _, _ = softmax_stream(c, X, W)
        L += pre + head
        L += iteration(c, ph, X, Y, U, W, reads_in_flight=AHEAD if nxt else 0, prefetch_next=((ph + 1) & 3) if nxt else None,
                       dma=pieces)
        L += check_and_count(c, "90f")
        L += ["s_cmp_ge_u32 %[t], %[tend]"]
        L += ["s_cbranch_scc1 2%df" % ((ph + 1) & 3)] if ph != 0 else ["s_cbranch_scc0 11b"]
    # ---- drains: PV of the last tile t - 1 (its P is the U of the phase that would come next); its first fragment reads were
    # issued by the iteration just left ----
    for ph in (1, 2, 3, 0):
        X, Y, U, W = roles[ph]
        L += ["2%d:" % ph]
        L += iteration(c, ph, X, Y, U, W, pv=True, softmax=False, qk=False, reads_in_flight=AHEAD if nxt else 0)
        L += ["s_mov_b32 %[code], 0", "s_branch 99f"]
    # ---- failed row-sum check in iteration t: its protocol, PV(t-1) and QK(t+1) are done, softmax(t) is not ----
    L += ["90:", "s_mov_b32 %[code], 1"]
    L += ["99:", "s_nop 15", "s_nop 15", "s_waitcnt lgkmcnt(0)"]
    return L


def write(c, path):
    lines = emit(c)
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_attn_q64.py -- do not edit.  Steady-state KV loop of the 64-queries-per-wave d = %d attention.\n" % c.d)
        f.write("#define ALG_%s_LOOP_ASM \\\n" % c.name)
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
        regs = ["a%d" % i for i in range(c.QA, c.AEND + (32 if EXP & {"ones", "mfma4"} else 0))] + ["v%d" % i for i in range(c.VB, c.VEND)] + \
               ["s%d" % i for i in range(KD, VD + 4)]
        f.write("#define ALG_%s_CLOBBERS \\\n  " % c.name + ", ".join('"%s"' % r for r in regs) + "\n")
        f.write("#define ALG_%s_O_OPERANDS(o) \\\n  " % c.name + ", ".join('[o%d] "+a"(o[%d])' % (i, i) for i in range(c.NO)) + "\n")
    return lines


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out128 = os.environ.get("ATTN128_Q64_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "attn128_q64_loop.inc")
    lines = write(Cfg(128, fma="nofma" not in EXP), out128)
    print("wrote", os.path.normpath(out128), len(lines), "lines,", sum(1 for l in lines if l.startswith("v_mfma")), "MFMAs")
    # (Cfg(64, fma=False) -- the same construction at half the head width -- still generates and still runs in the emulator suite; its
    # kernel, ALG_ATTN_PP=6, measured 1.5-2 % slower than the 8-wave statement in the model for two rounds and left the library in round 6)


if __name__ == "__main__":
    main()
