#!/usr/bin/env python3
"""Generates alg_amd/csrc/attn_pipe_loop.inc: the steady-state KV loop of the d = 64 attention as ONE inline-asm statement.

Why (profiles/r3_attention_d64_mix_microbench.txt): on gfx950 matrix and vector work of DIFFERENT waves of a SIMD do not
overlap, vector work issued by the SAME wave right behind its own MFMA does -- 1 MFMA + {2 v_exp, 1 v_cvt_pk_bf16_f32, 2 v_add,
1 ds_read_b128} costs 18.3 ns of SIMD time against 16.3 ns for the MFMA alone, while the straight loop of attention.hip (QK ->
softmax -> PV, each phase on its own) spends 28.6 ns per MFMA.  (v_dot2c_f32_bf16 does NOT hide: +7 ns per MFMA; plain adds do.)

The loop is software-pipelined over KV tiles of 64: iteration t issues, per wave (32 queries; a workgroup is four waves = 128 queries, two workgroups per CU),
    PV(t-1): O^T += V(t-1)^T P(t-1)^T      8 MFMAs, A = V^T fragments from LDS, B = P(t-1) (registers), C/D = a[0:31]
    QK(t+1): S(t+1)^T = K(t+1) Q^T         8 MFMAs, A = K fragments from LDS,   B = Q (registers), first k-step from C = 0
    softmax(t): P(t) = bf16(exp2(S(t))), row sum     16 score pairs, ONE PAIR PER MFMA GAP: exp, exp, cvt_pk (of the previous
                                                     pair), add, add -- none of it depends on this iteration's MFMAs
Scores arrive in log2 units with the running offset at zero (attention.hip, softmax_tile_zero's common path): the statement is
entered only by waves whose offsets are all zero and leaves as soon as a tile's row sum leaves [0, 2^80) -- the exact path
(tile max, rescale) stays in C++.

Collective protocol (identical in the C++ loop around the statement, so waves of one workgroup may be in either): at the top
of iteration t   s_waitcnt vmcnt(4); s_barrier;  DMA K(t+3) -> K slot (t+3) & 3, V^T(t+2) -> V slot (t+2) & 3   (2 KiB = two
instructions per wave and tile each; the counted wait leaves the previous iteration's four in flight: two tiles of prefetch).
Iteration t reads K(t+1), V(t-1) (pipelined form) or K(t), V(t) (straight form): all resident in the 4-slot rings.

Unrolled x 4 (ring slot = t & 3 as immediates; the S / P register roles alternate with t & 1).  Entered at t = 1 (mod 4) through
a warm-up (QK(t) on its own; no PV: the caller has finished tile t-1), runs whole groups of four while t + 4 <= t_end, then
drains PV of its last tile.  Register plan (named literally, clobbered):
    v[64:95]  SA   v[96:127] SB     score tiles (sub-tile 0: +0..15, sub-tile 1: +16..31); roles alternate
    v[128:143] PA  v[144:159] PB    packed probabilities: register n = pair (S[2n], S[2n+1])
    v160 tile sum, v[161:164] exp results in flight, v165 scratch
    a[0:31]   O^T accumulators (two 32x32 tiles)
    a[32:63]  eight fragment buffers (ring, one per MFMA, read four MFMAs ahead; ds_read_b128 straight into AccVGPRs)
    a[64:79]  Q fragments (loaded by the statement: four 16-byte pieces per lane)
(102 ArchVGPRs + 80 AccVGPRs: with hipcc's own registers the kernel stays at 256 per lane = two waves per SIMD)
Operands: o0..o31 "+v" (O^T elements, moved to / from a[0:31]), l "+v" running row sum, t "+s" iteration index (in: first
iteration, = 1 mod 4; out: the iteration the caller continues with), code "=s" (0: t's top-of-iteration protocol NOT done,
caller continues normally; 1: row-sum check failed in iteration t -- its protocol, PV(t-1) and QK(t+1) are done, softmax(t)
is not: the caller redoes tile t from QK), lk0..lk3 / lv0..lv3 "v" LDS byte address of the lane's K / V^T fragment per
k-step (ring base included; slot and sub-tile are immediates), kvo0/1, vvo0/1 "+v" DMA byte offsets of this lane (rows srow,
srow + 32) into the K / V^T panels AT TILE (t + 3) / (t + 2) of the entry iteration (advanced inside), qvo "v" byte offset of the lane's Q row, kb / vb /
qb "s" 64-bit panel bases, kstep "s" bytes per K tile, tend "s" (whole groups of four run while t + 4 <= tend; the caller guarantees t + 3 < T inside),
wk / wv "s" = LDS byte address of the K / V^T ring + wave * 1024 (DMA destination of this wave).
"""
import os

OACC, FR, Q = 0, 32, 64        # AccVGPRs: O^T tiles, fragment ring, Q fragments
WAIT_PAIRS = os.environ.get("ATTN_PIPE_WAIT_PAIRS", "0") == "1"   # experiment knob (round 5): one fragment wait per two MFMAs
NO_NOP = os.environ.get("ATTN_PIPE_NO_NOP", "0") == "1"           # experiment knob (round 5): no s_nop between an M0 write and its LDS-DMA
SUM16 = os.environ.get("ATTN_PIPE_SUM16", "0") == "1"             # experiment knob (round 5, TIMING ONLY -- wrong row sums): row sums on the matrix pipe
LACC, ONES = 80, 84            # (SUM16) AccVGPRs: 16x16 row-sum accumulator, the selector A operand
NW = 4                         # waves per workgroup (configure())


def configure(nw):
    """4-wave form: ArchVGPRs v[64:165], O exchanged through 32 "+v" operands, two DMA pieces per wave, tile and operand.
    8-wave form (one 256-query unit per workgroup, half the L2 -> LDS traffic per MFMA): hipcc grants such a workgroup 128 + 128
    registers per lane, so the statement's ArchVGPRs are v[26:127] and O travels in 32 "+a" operands (v_accvgpr_mov)."""
    global SA, SB, PA, PB, TS, E0, E1, E2, E3, SCR, NW, VBASE
    NW = nw
    VBASE = 64 if nw == 4 else 26
    SA, SB, PA, PB = VBASE, VBASE + 32, VBASE + 64, VBASE + 80
    TS, E0, E1, E2, E3, SCR = (VBASE + 96 + i for i in range(6))


configure(4)
TILE = 8192

v = lambda i: "v%d" % i
vr = lambda i, n: "v[%d:%d]" % (i, i + n - 1)
ar = lambda i, n: "a[%d:%d]" % (i, i + n - 1)


def frag_read(buf, which, slot, half, kstep):
    """ds_read_b128 of one fragment: K (which = 'k') sub-tile `half` k-step `kstep`, or V^T d-tile `half` kv block `kstep`"""
    # address = lane part(kstep) (carries the ring base) + slot * TILE + half * 4096
    return "ds_read_b128 %s, %%[l%s%d] offset:%d" % (ar(FR + 4 * buf, 4), which, kstep, slot * TILE + half * 4096)


def softmax_gap(S, P, n, first, last_of_tile=False):
    """VALU work of one MFMA gap: exp2 of pair n (if n < 16), pack + row sum of pair n - 1 (if n >= 1).
    Pair n lives in E0/E1 (n even) or E2/E3 (n odd) until it is packed in the next gap."""
    out = []
    ea, eb = (E0, E1) if n % 2 == 0 else (E2, E3)
    pa, pb = (E2, E3) if n % 2 == 0 else (E0, E1)   # previous pair
    if n >= 1:
        out.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P + n - 1), v(pa), v(pb)))
        if n - 1 == 0:
            out.append("v_add_f32 %s, %s, %s" % (v(TS), v(pa), v(pb)))
        elif SUM16:
            pass
        else:
            out.append("v_add_f32 %s, %s, %s" % (v(SCR), v(pa), v(pb)))
            out.append("v_add_f32 %s, %s, %s" % (v(TS), v(TS), v(SCR)))
    if n < 16:
        out.append("v_exp_f32 %s, %s" % (v(ea), v(S + 2 * n)))
        out.append("v_exp_f32 %s, %s" % (v(eb), v(S + 2 * n + 1)))
    return out


def top_protocol(phase, spread=False):
    """top of iteration t (t & 3 == phase): all but the previous iteration's four DMAs have landed (K(t+1), V(t) and older),
    everybody is done with iteration t - 1; then this wave's share of K(t+3) and V^T(t+2) -- two tiles ahead, into the slots of
    K(t-1) / V^T(t-2): two 1 KiB pieces each (four waves stage an 8 KiB tile).
    spread: returns (head, groups) -- the four DMAs go out one per MFMA gap behind the barrier instead of as a block"""
    ks, vs = (phase + 3) & 3, (phase + 2) & 3
    rounds = 8 // NW
    head = ["s_waitcnt vmcnt(%d)" % (2 * rounds), "s_barrier"]
    groups = []
    for r in range(rounds):
        groups.append(["s_add_u32 m0, %%[wk], %d" % (ks * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[kvo%d], %%[kb]" % r, "v_add_u32 %%[kvo%d], %%[kstep], %%[kvo%d]" % (r, r)])
    for r in range(rounds):
        groups.append(["s_add_u32 m0, %%[wv], %d" % (vs * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[vvo%d], %%[vb]" % r, "v_add_u32 %%[vvo%d], 0x80, %%[vvo%d]" % (r, r)])
    if spread:
        return head, groups
    return head + [ln for g in groups for ln in g]


def first_reads(phase):
    """the first four fragment reads of a full iteration at `phase`: V^T(t-1) kv blocks 0, 1 for both d-tiles.  They do not
    depend on the iteration's barrier (V^T(t-1) landed two iterations earlier), so the PREVIOUS iteration issues them behind its
    twelfth MFMA (buffers 0-3 are free from there on) and the LDS latency hides under the top-of-iteration wait + barrier."""
    vslot = (phase - 1) & 3
    return [frag_read(j, "v", vslot, half, kstep) for j, (half, kstep) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])]


def iteration(phase, X, Y, U, W, pv=True, softmax=True, qk=True, reads_in_flight=False, prefetch_next=None, dma_groups=None,
              pre_exp=False):
    """one pipelined iteration at ring phase t & 3 == phase.  MFMA stream (16, or 8 without PV): the four PV MFMAs of kv blocks
    0, 1 first (their V^T fragments do not depend on this iteration's barrier), then QK and PV alternating, QK last."""
    kslot, vslot = (phase + 1) & 3, (phase - 1) & 3
    mf = []   # (kind, half, kstep)
    if pv:
        mf += [("v", 0, 0), ("v", 1, 0), ("v", 0, 1), ("v", 1, 1)]
        rest_v = [("v", 0, 2), ("v", 1, 2), ("v", 0, 3), ("v", 1, 3)]
    else:
        rest_v = []
    qks = [("k", s, ks) for ks in range(4) for s in range(2)] if qk else []
    # alternate: QK, PV, QK, PV, ... then the remaining QKs
    i = 0
    while rest_v or qks[i:]:
        if qks[i:]:
            mf.append(qks[i]); i += 1
        if rest_v:
            mf.append(rest_v.pop(0))
    n_m = len(mf)
    lines = []
    AHEAD = 4
    def read(j):
        kind, half, kstep = mf[j]
        return frag_read(j % 8, kind, kslot if kind == "k" else vslot, half, kstep)
    if not reads_in_flight:
        for j in range(min(AHEAD, n_m)):
            lines.append(read(j))
    pair = 1 if (softmax and pre_exp) else 0     # pre_exp: the caller has issued the two exps of pair 0 (softmax_pre)
    n_pairs = 16 if softmax else 0
    # pairs per gap: spread 17 VALU groups (16 exps + the trailing pack) over the gaps
    per_gap = max(1, 16 // n_m) if softmax else 0
    seen_first = {"k0": False, "k1": False}
    for j, (kind, half, kstep) in enumerate(mf):
        if WAIT_PAIRS:
            # ONE counted wait per TWO fragments (round 5: every s_waitcnt is an issue slot of a loop that is bound by issue slots):
            # in front of an even MFMA j the reads up to j + AHEAD - 1 have been issued; fragments j and j + 1 must have arrived
            if j % 2 == 0:
                last = (n_m + AHEAD - 1) if prefetch_next is not None else (n_m - 1)
                lines.append("s_waitcnt lgkmcnt(%d)" % max(0, min(j + AHEAD - 1, last) - (j + 1)))
        else:
            outstanding = (AHEAD if prefetch_next is not None else min(AHEAD, n_m - j)) - 1   # reads issued after read j
            lines.append("s_waitcnt lgkmcnt(%d)" % outstanding)
        fr = ar(FR + 4 * (j % 8), 4)
        if kind == "k":
            acc = vr(Y + 16 * half, 16)
            c = acc if seen_first["k%d" % half] else "0"
            seen_first["k%d" % half] = True
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fr, ar(Q + 4 * kstep, 4), c))
        else:
            acc = ar(16 * half, 16)
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fr, vr(U + 4 * kstep, 4), acc))
            if SUM16 and half == 1:
                # one 4-pass MFMA per kv block: D[0][n] / D[1][n] = sum over the block's 16 keys of P for queries n / 16 + n
                lines.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (ar(LACC, 4), ar(ONES, 4), vr(U + 4 * kstep, 4), ar(LACC, 4)))
        if j + AHEAD < n_m:
            lines.append(read(j + AHEAD))
        elif prefetch_next is not None and j + AHEAD - n_m < AHEAD:
            lines.append(first_reads(prefetch_next)[j + AHEAD - n_m])   # next iteration's reads 0..3 behind MFMAs 12..15
        tail_dma = []
        if dma_groups and j < len(dma_groups):
            if NO_NOP and softmax and pair <= 16:
                # the M0 write, then this gap's softmax slice, then the LDS-DMA that reads M0: real work instead of the s_nop
                lines.append(dma_groups[j][0])
                tail_dma = [ln for ln in dma_groups[j][1:] if not ln.startswith("s_nop")]
            else:
                lines += dma_groups[j]                                   # one DMA per gap behind the barrier
        for g in range(per_gap):
            if softmax and pair <= 16:
                if g > 0:
                    lines.append("s_nop 1")     # the pack below reads what the two exps just above wrote (trans -> VALU use)
                lines += softmax_gap(X, W, pair, pair == 0)
                pair += 1
        lines += tail_dma
    while softmax and pair <= 16:
        lines.append("s_nop 1")
        lines += softmax_gap(X, W, pair, False)
        pair += 1
    return lines


def softmax_pre(S):
    """the two exps of score pair 0, issued AHEAD of the top-of-iteration wait + barrier (the scores are complete)"""
    return ["v_exp_f32 %s, %s" % (v(E0), v(S)), "v_exp_f32 %s, %s" % (v(E1), v(S + 1))]


def check_and_count(fail_label):
    """row-sum check of the iteration just issued, then t += 1"""
    return ["v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(TS), "s_nop 4", "s_cbranch_vccnz %s" % fail_label,   # !(2^80 > sum)
            "v_add_f32 %%[l], %%[l], %s" % v(TS), "s_add_u32 %[t], %[t], 1"]


def emit():
    L = []
    # ---- entry: O -> a[0:31], Q fragments, constants ----
    L += [("v_accvgpr_write_b32 a%d, %%[o%d]" if NW == 4 else "v_accvgpr_mov_b32 a%d, %%[o%d]") % (i, i) for i in range(32)]
    L += ["global_load_dwordx4 %s, %%[qvo], %%[qb] offset:%d" % (ar(Q + 4 * ks, 4), 32 * ks) for ks in range(4)]
    roles = {1: (SA, SB, PA, PB), 2: (SB, SA, PB, PA), 3: (SA, SB, PA, PB), 0: (SB, SA, PB, PA)}
    # ---- warm-up at phase 1: top protocol, QK(t) alone into X = SA (K(t) sits in slot 1 = the "next" slot of phase 0) ----
    if SUM16:
        L += ["v_accvgpr_write_b32 a%d, 0" % (LACC + i) for i in range(4)] + ["v_mov_b32 %s, 0x3f803f80" % v(SCR)] + ["v_accvgpr_write_b32 a%d, %s" % (ONES + i, v(SCR)) for i in range(4)]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]   # Q (and, once, whatever the caller had in flight)
    L += top_protocol(1)
    X, Y, U, W = roles[1]
    L += iteration(0, Y, X, U, W, pv=False, softmax=False)      # phase 0's "next" K slot is slot 1: S(t) -> SA
    L += ["s_nop 15", "s_nop 15"]                                  # S(t) complete before the first exp reads it
    L += iteration(1, X, Y, U, W, pv=False, prefetch_next=2)      # QK(t+1) -> SB under softmax(t) -> PB
    L += check_and_count("90f")
    L += ["s_branch 12f"]
    # ---- the loop: phases 1, 2, 3, 0 ----
    L += ["11:"]
    for ph in (1, 2, 3, 0):
        if ph == 2:
            L += ["12:"]
        X, Y, U, W = roles[ph]
        head, groups = top_protocol(ph, spread=True)
        L += softmax_pre(X) + head
        L += iteration(ph, X, Y, U, W, reads_in_flight=True, prefetch_next=(ph + 1) & 3, dma_groups=groups, pre_exp=True)
        L += check_and_count("90f")
    # after phase 0: t = 1 (mod 4) again.  Another whole group?  (t + 4 <= tend)
    L += ["s_add_u32 %[code], %[t], 4", "s_cmp_le_u32 %[code], %[tend]", "s_cbranch_scc1 11b"]
    # ---- drain: PV of the last tile (P in PB after phase 0: W of phase 0 = PA? roles[0] = (SB, SA, PB, PA): W = PA) ----
    X, Y, U, W = roles[0]
    L += iteration(1, Y, X, W, U, pv=True, softmax=False, qk=False, reads_in_flight=True)   # V slot (1 - 1) & 3 = slot of tile t - 1
    L += ["s_mov_b32 %[code], 0", "s_branch 99f"]
    # ---- failed row-sum check in iteration t: its MFMAs are issued; leave with code 1 ----
    L += ["90:", "s_mov_b32 %[code], 1"]
    L += ["99:", "s_nop 15", "s_nop 15"]
    L += [("v_accvgpr_read_b32 %%[o%d], a%d" if NW == 4 else "v_accvgpr_mov_b32 %%[o%d], a%d") % (i, i) for i in range(32)]
    L += ["s_waitcnt lgkmcnt(0)"]
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("ATTN_PIPE_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "attn_pipe_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_attn_pipe.py -- do not edit.  Steady-state KV loop of the pipelined d = 64 attention.\n")
        for nw, tag in ((4, ""), (8, "8")):
            configure(nw)
            lines = emit()
            f.write("#define ALG_ATTN_PIPE%s_LOOP_ASM \\\n" % tag)
            for ln in lines:
                f.write('  "%s\\n\\t" \\\n' % ln)
            f.write('  ""\n')
            regs = ["a%d" % i for i in range(88 if SUM16 else 80)] + ["v%d" % i for i in range(VBASE, VBASE + 102)]
            f.write("#define ALG_ATTN_PIPE%s_CLOBBERS \\\n  " % tag + ", ".join('"%s"' % r for r in regs) + '\n')
            f.write("#define ALG_ATTN_PIPE%s_O_OPERANDS(o) \\\n  " % tag +
                    ", ".join('[o%d] "+%s"(o[%d])' % (i, "v" if nw == 4 else "a", i) for i in range(32)) + '\n')
            print("wrote", os.path.normpath(path), "form", nw, len(lines), "lines,", sum(1 for l in lines if l.startswith("v_mfma")), "MFMAs")
    configure(4)


if __name__ == "__main__":
    main()
