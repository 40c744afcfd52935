#!/usr/bin/env python3
"""Generates alg_amd/csrc/attn_pipe64_loop.inc: the steady-state KV loop of the d = 64 attention with 64 QUERIES PER WAVE as one
inline-asm statement -- the sibling of gen_attn_pipe.py (read its header for the why and for the collective protocol, ring
layout, entry / exit conventions, which are the same).  What differs: a wave owns two 32-query halves, every K / V^T fragment
read from LDS feeds TWO MFMAs (one per half), so the fragment traffic per MFMA is half that of the 32-query forms -- the kernels
are limited by energy per FLOP, not by issue slots.  Four waves = one 256-query unit per workgroup, ONE wave
per SIMD (199 ArchVGPRs + 128 AccVGPRs named here + 64 AccVGPR operands).

Per wave and 64-key tile, iteration t issues 32 MFMAs -- PV(t-1): 2 d-tiles x 4 kv blocks x 2 halves, QK(t+1): 2 sub-tiles x
4 k-steps x 2 halves --, 16 fragment reads, four DMAs (two 1 KiB pieces of K(t+3), two of V^T(t+2)) and softmax(t): 32 score
pairs (16 per half), one per MFMA gap: exp, exp, then in the next gap cvt_pk, add, add.
Registers (named literally, clobbered):
    v[56:119] SA, v[120:183] SB      scores: half h at +32 h (sub-tile 0: +0..15, sub-tile 1: +16..31); roles alternate with t & 1
    v[184:215] PA, v[216:247] PB     packed probabilities: half h at +16 h, register n = pair (S[2n], S[2n+1])
    v248, v249 tile sums of the halves, v[250:253] exp results in flight, v254 scratch
    a[0:63]   O^T: half h, d-tile d at a[32 h + 16 d ..]      a[64:95] eight fragment buffers      a[96:127] Q: half h, k-step k at 96 + 16 h + 4 k
Operands: o0..o63 "+a", l0 / l1 "+v" running row sums of the halves, t "+s", code "=&s", lk0..3 / lv0..3 "v", kvo0/1, vvo0/1 "+v",
qvo0 / qvo1 "v" byte offsets of the lane's two Q rows, kb / vb / qb "s" 64-bit bases, kstep, tend, wk, wv "s".
"""
import os

SA, SB, PA, PB = 56, 120, 184, 216
TS0, TS1, E0, E1, E2, E3, SCR = range(248, 255)
OACC, FR, Q = 0, 64, 96
TILE = 8192
AHEAD = 4                      # fragments in flight ahead of their first MFMA

v = lambda i: "v%d" % i
vr = lambda i, n: "v[%d:%d]" % (i, i + n - 1)
ar = lambda i, n: "a[%d:%d]" % (i, i + n - 1)


def frag_read(buf, which, slot, half, step):
    return "ds_read_b128 %s, %%[l%s%d] offset:%d" % (ar(FR + 4 * buf, 4), which, step, slot * TILE + half * 4096)


def valu_groups(S, P):
    """softmax(t) as 33 groups, one per MFMA gap: group n = {pack + row sum of pair n - 1, exp, exp of pair n}; group 32 = pack + sum of
    pair 31.  Pair n: half n >> 4, scores S[32 h + 2 i], S[32 h + 2 i + 1] (i = n & 15) -> P[16 h + i]."""
    def fin(n):
        h, i = n >> 4, n & 15
        ea, eb = (E0, E1) if n % 2 == 0 else (E2, E3)
        ts = TS0 if h == 0 else TS1
        out = ["v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P + 16 * h + i), v(ea), v(eb))]
        if i == 0:
            out.append("v_add_f32 %s, %s, %s" % (v(ts), v(ea), v(eb)))
        else:
            out += ["v_add_f32 %s, %s, %s" % (v(SCR), v(ea), v(eb)), "v_add_f32 %s, %s, %s" % (v(ts), v(ts), v(SCR))]
        return out
    groups = []
    for n in range(32):
        h, i = n >> 4, n & 15
        ea, eb = (E0, E1) if n % 2 == 0 else (E2, E3)
        g = fin(n - 1) if n >= 1 else []
        g += ["v_exp_f32 %s, %s" % (v(ea), v(S + 32 * h + 2 * i)), "v_exp_f32 %s, %s" % (v(eb), v(S + 32 * h + 2 * i + 1))]
        groups.append(g)
    groups.append(fin(31))
    return groups


def top_protocol(phase):
    ks, vs = (phase + 3) & 3, (phase + 2) & 3
    head = ["s_waitcnt vmcnt(4)", "s_barrier"]
    groups = []
    for r in range(2):
        groups.append(["s_add_u32 m0, %%[wk], %d" % (ks * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[kvo%d], %%[kb]" % r, "v_add_u32 %%[kvo%d], %%[kstep], %%[kvo%d]" % (r, r)])
    for r in range(2):
        groups.append(["s_add_u32 m0, %%[wv], %d" % (vs * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[vvo%d], %%[vb]" % r, "v_add_u32 %%[vvo%d], 0x80, %%[vvo%d]" % (r, r)])
    return head, groups


def frag_order(pv, qk):
    """fragment stream: V^T kv blocks 0, 1 (both d-tiles) first -- prefetched across the barrier --, then K and V^T alternating"""
    fr = []
    rest_v = []
    if pv:
        fr += [("v", 0, 0), ("v", 1, 0), ("v", 0, 1), ("v", 1, 1)]
        rest_v = [("v", 0, 2), ("v", 1, 2), ("v", 0, 3), ("v", 1, 3)]
    ks = [("k", s, k) for k in range(4) for s in range(2)] if qk else []
    i = 0
    while rest_v or ks[i:]:
        if ks[i:]:
            fr.append(ks[i]); i += 1
        if rest_v:
            fr.append(rest_v.pop(0))
    return fr


def first_reads(phase):
    vslot = (phase - 1) & 3
    return [frag_read(j, "v", vslot, half, kk) for j, (half, kk) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])]


def iteration(phase, X, Y, U, W, pv=True, softmax=True, qk=True, reads_in_flight=False, prefetch_next=None, dma_groups=None,
              pre_group=False):
    kslot, vslot = (phase + 1) & 3, (phase - 1) & 3
    fr = frag_order(pv, qk)
    n_f = len(fr)
    lines = []
    def read(f):
        kind, half, step = fr[f]
        return frag_read(f % 8, kind, kslot if kind == "k" else vslot, half, step)
    if not reads_in_flight:
        for f in range(min(AHEAD, n_f)):
            lines.append(read(f))
    groups = valu_groups(X, W) if softmax else []
    gi = 1 if (softmax and pre_group) else 0
    per_gap = 1 if n_f >= 16 else 2                   # warm-up (16 MFMAs): two groups per gap
    seen_first = set()
    gap = 0
    for f, (kind, half, step) in enumerate(fr):
        outstanding = (AHEAD if prefetch_next is not None else min(AHEAD, n_f - f)) - 1
        lines.append("s_waitcnt lgkmcnt(%d)" % outstanding)
        fa = ar(FR + 4 * (f % 8), 4)
        for h in range(2):
            if kind == "k":
                acc = vr(Y + 32 * h + 16 * half, 16)
                c = acc if (h, half) in seen_first else "0"
                seen_first.add((h, half))
                lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fa, ar(Q + 16 * h + 4 * step, 4), c))
            else:
                acc = ar(OACC + 32 * h + 16 * half, 16)
                lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fa, vr(U + 16 * h + 4 * step, 4), acc))
            if h == 1:      # behind the fragment's second MFMA: the read four fragments ahead (its buffer's last reader is older)
                if f + AHEAD < n_f:
                    lines.append(read(f + AHEAD))
                elif prefetch_next is not None and f + AHEAD - n_f < AHEAD:
                    lines.append(first_reads(prefetch_next)[f + AHEAD - n_f])
            if dma_groups and gap < len(dma_groups):
                lines += dma_groups[gap]
            for k in range(per_gap):
                if gi < len(groups):
                    if k > 0:
                        lines.append("s_nop 1")
                    lines += groups[gi]
                    gi += 1
            gap += 1
    while gi < len(groups):
        lines.append("s_nop 1")
        lines += groups[gi]
        gi += 1
    return lines


def check_and_count(fail_label):
    return ["v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(TS0), "s_nop 4", "s_cbranch_vccnz %s" % fail_label,   # !(2^80 > sum)
            "v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(TS1), "s_nop 4", "s_cbranch_vccnz %s" % fail_label,
            "v_add_f32 %%[l0], %%[l0], %s" % v(TS0), "v_add_f32 %%[l1], %%[l1], %s" % v(TS1), "s_add_u32 %[t], %[t], 1"]


def emit():
    L = []
    L += ["v_accvgpr_mov_b32 a%d, %%[o%d]" % (i, i) for i in range(64)]
    for h in range(2):
        L += ["global_load_dwordx4 %s, %%[qvo%d], %%[qb] offset:%d" % (ar(Q + 16 * h + 4 * ks, 4), h, 32 * ks) for ks in range(4)]
    roles = {1: (SA, SB, PA, PB), 2: (SB, SA, PB, PA), 3: (SA, SB, PA, PB), 0: (SB, SA, PB, PA)}
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    head, groups = top_protocol(1)
    L += head + [ln for g in groups for ln in g]
    X, Y, U, W = roles[1]
    L += iteration(0, Y, X, U, W, pv=False, softmax=False)         # QK(t) alone -> SA (K(t) sits in slot 1)
    L += ["s_nop 15", "s_nop 15"]
    L += iteration(1, X, Y, U, W, pv=False, prefetch_next=2)        # QK(t+1) -> SB under softmax(t) -> PB
    L += check_and_count("90f")
    L += ["s_branch 12f"]
    L += ["11:"]
    for ph in (1, 2, 3, 0):
        if ph == 2:
            L += ["12:"]
        X, Y, U, W = roles[ph]
        head, groups = top_protocol(ph)
        L += valu_groups(X, W)[0] + head
        L += iteration(ph, X, Y, U, W, reads_in_flight=True, prefetch_next=(ph + 1) & 3, dma_groups=groups, pre_group=True)
        L += check_and_count("90f")
    L += ["s_add_u32 %[code], %[t], 4", "s_cmp_le_u32 %[code], %[tend]", "s_cbranch_scc1 11b"]
    X, Y, U, W = roles[0]
    L += iteration(1, Y, X, W, U, pv=True, softmax=False, qk=False, reads_in_flight=True)    # drain: PV of the last tile
    L += ["s_mov_b32 %[code], 0", "s_branch 99f"]
    L += ["90:", "s_mov_b32 %[code], 1"]
    L += ["99:", "s_nop 15", "s_nop 15"]
    L += ["v_accvgpr_mov_b32 %%[o%d], a%d" % (i, i) for i in range(64)]
    L += ["s_waitcnt lgkmcnt(0)"]
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("ATTN_PIPE64_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "attn_pipe64_loop.inc")
    lines = emit()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_attn_pipe64.py -- do not edit.  Steady-state KV loop of the 64-query pipelined d = 64 attention.\n")
        f.write("#define ALG_ATTN_PIPE64_LOOP_ASM \\\n")
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
        regs = ["a%d" % i for i in range(128)] + ["v%d" % i for i in range(56, 255)]
        f.write("#define ALG_ATTN_PIPE64_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
        f.write("#define ALG_ATTN_PIPE64_O_OPERANDS(o) \\\n  " + ", ".join('[o%d] "+a"(o[%d])' % (i, i) for i in range(64)) + '\n')
    print("wrote", os.path.normpath(path), len(lines), "lines,", sum(1 for l in lines if l.startswith("v_mfma")), "MFMAs")


if __name__ == "__main__":
    main()
