#!/usr/bin/env python3
"""Generates alg_amd/csrc/attn128_pipe_loop.inc: the steady-state KV loop of the d = 128 self-attention (Wan / HunyuanVideo DiTs)
as ONE inline-asm statement -- the d = 128 sibling of gen_attn_pipe.py (read its header for the why: vector work overlaps matrix
work only when the SAME wave issues it right behind its own MFMA).

Per wave (32 queries; a workgroup is four waves, one per SIMD) and 64-key tile, iteration t issues
    PV(t-1): O^T += V(t-1)^T P(t-1)^T    16 MFMAs (four 32-row d-tiles x four kv blocks), C/D = a[0:63]
    QK(t+1): S(t+1)^T = K(t+1) Q^T       16 MFMAs (two 32-key sub-tiles x eight k-steps), B = Q in a[96:127]
    softmax(t): p = exp2(s * c - m * c) against the CURRENT running max m (attention128.hip's lazy form: the statement bails
                out when a tile's row sum leaves [0, 2^80), the exact path stays in C++): per score pair 2 fma, 2 exp2, 1 pack,
                2 adds -- one pair per TWO MFMA gaps
K tile = 64 keys x 256 B (swizzle: 16-byte slot ^ (row & 15)), V^T tile = 128 d-rows x 128 B (swizzle (row >> 1) & 7): 16 KiB
each, four-slot rings (128 KiB of LDS), two tiles of DMA prefetch, counted vmcnt(8) (eight 1 KiB pieces per wave and
iteration: four of K, four of V^T).  Collective protocol, register roles, entry / exit conventions as in gen_attn_pipe.py.
Registers: v[64:95] SA, v[96:127] SB, v[128:143] PA, v[144:159] PB, v160 tile sum, v[161:164] exp results, v165 scratch,
v[166:169] fma results; a[0:63] O^T, a[64:95] fragment ring (eight buffers), a[96:127] Q fragments.
Operands: o0..o63 "+v", l "+v", negmc "v" (-m * c of the lane's query), c "s", t "+s", code "=&s", lk0..lk7 / lv0..lv3 "v"
fragment addresses per k-step / kv block, kvo0..3 / vvo0..3 "+v" DMA byte offsets (tile t + 3 / t + 2 at entry), qvo "v",
kb / vb / qb "s" 64-bit bases, kstep "s" bytes per K tile, tend "s", wk / wv "s" ring address + wave * 1024.
"""
import os

SA, SB, PA, PB = 64, 96, 128, 144
TS, E0, E1, E2, E3, SCR, F0, F1, F2, F3 = range(160, 170)
OACC, FR, Q = 0, 64, 96        # AccVGPRs
TILE = 16384                   # K tile = V^T tile
AHEAD = 4                      # fragment reads in flight ahead of their MFMA

v = lambda i: "v%d" % i
vr = lambda i, n: "v[%d:%d]" % (i, i + n - 1)
ar = lambda i, n: "a[%d:%d]" % (i, i + n - 1)


def frag_read(buf, which, slot, half, step):
    """ds_read_b128 of one fragment: K sub-tile `half` (0, 1: +8192) k-step `step` (0..7), or V^T d-tile `half` (0..3: +4096) kv
    block `step` (0..3).  The lane part (operand) carries the ring base and the swizzled 16-byte slot of the step."""
    off = slot * TILE + half * (8192 if which == "k" else 4096)
    return "ds_read_b128 %s, %%[l%s%d] offset:%d" % (ar(FR + 4 * buf, 4), which, step, off)


def valu_groups(S, P):
    """softmax(t) as 33 small groups, one per MFMA gap (group g goes behind MFMA g - 1; group 0 in front of the barrier):
    pair n: group 2n = {pack + row sum of pair n - 1, fma, fma}, group 2n + 1 = {exp, exp}; group 32 = pack + sum of pair 15"""
    groups = []
    def fin(n):     # pack + row sum of pair n
        ea, eb = (E0, E1) if n % 2 == 0 else (E2, E3)
        out = ["v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P + n), v(ea), v(eb))]
        if n == 0:
            out.append("v_add_f32 %s, %s, %s" % (v(TS), v(ea), v(eb)))
        else:
            out += ["v_add_f32 %s, %s, %s" % (v(SCR), v(ea), v(eb)), "v_add_f32 %s, %s, %s" % (v(TS), v(TS), v(SCR))]
        return out
    for n in range(16):
        fa, fb = (F0, F1) if n % 2 == 0 else (F2, F3)
        ea, eb = (E0, E1) if n % 2 == 0 else (E2, E3)
        g = fin(n - 1) if n >= 1 else []
        g += ["v_fma_f32 %s, %s, %%[c], %%[negmc]" % (v(fa), v(S + 2 * n)), "v_fma_f32 %s, %s, %%[c], %%[negmc]" % (v(fb), v(S + 2 * n + 1))]
        groups.append(g)
        groups.append(["v_exp_f32 %s, %s" % (v(ea), v(fa)), "v_exp_f32 %s, %s" % (v(eb), v(fb))])
    groups.append(fin(15))
    return groups


def top_protocol(phase):
    """(head, DMA groups) of iteration t, t & 3 == phase: all but the previous iteration's eight DMAs have landed; behind the
    barrier this wave's four 1 KiB pieces of K(t+3) and four of V^T(t+2), one per MFMA gap"""
    ks, vs = (phase + 3) & 3, (phase + 2) & 3
    head = ["s_waitcnt vmcnt(8)", "s_barrier"]
    groups = []
    for r in range(4):
        groups.append(["s_add_u32 m0, %%[wk], %d" % (ks * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[kvo%d], %%[kb]" % r, "v_add_u32 %%[kvo%d], %%[kstep], %%[kvo%d]" % (r, r)])
    for r in range(4):
        groups.append(["s_add_u32 m0, %%[wv], %d" % (vs * TILE + r * 4096), "s_nop 0",
                       "global_load_lds_dwordx4 %%[vvo%d], %%[vb]" % r, "v_add_u32 %%[vvo%d], 0x80, %%[vvo%d]" % (r, r)])
    return head, groups


def mfma_order(pv, qk):
    """MFMA stream: PV kv block 0 (four d-tiles) first -- its fragments are prefetched across the barrier --, then QK and PV
    alternating, the remaining QKs last"""
    mf = []
    rest_v = []
    if pv:
        mf += [("v", dt, 0) for dt in range(4)]
        rest_v = [("v", dt, kk) for kk in range(1, 4) for dt in range(4)]
    qks = [("k", s, ks) for ks in range(8) for s in range(2)] if qk else []
    i = 0
    while rest_v or qks[i:]:
        if qks[i:]:
            mf.append(qks[i]); i += 1
        if rest_v:
            mf.append(rest_v.pop(0))
    return mf


def first_reads(phase):
    vslot = (phase - 1) & 3
    return [frag_read(j, "v", vslot, dt, 0) for j, dt in enumerate(range(4))]


def iteration(phase, X, Y, U, W, pv=True, softmax=True, qk=True, reads_in_flight=False, prefetch_next=None, dma_groups=None,
              pre_group=False):
    kslot, vslot = (phase + 1) & 3, (phase - 1) & 3
    mf = mfma_order(pv, qk)
    n_m = len(mf)
    lines = []
    def read(j):
        kind, half, step = mf[j]
        return frag_read(j % 8, kind, kslot if kind == "k" else vslot, half, step)
    if not reads_in_flight:
        for j in range(min(AHEAD, n_m)):
            lines.append(read(j))
    groups = valu_groups(X, W) if softmax else []
    gi = 1 if (softmax and pre_group) else 0          # pre_group: group 0 was issued in front of the barrier
    per_gap = 1 if n_m >= 32 else 2                   # warm-up (16 MFMAs): two groups per gap
    seen_first = {0: False, 1: False}
    for j, (kind, half, step) in enumerate(mf):
        outstanding = (AHEAD if prefetch_next is not None else min(AHEAD, n_m - j)) - 1
        lines.append("s_waitcnt lgkmcnt(%d)" % outstanding)
        fr = ar(FR + 4 * (j % 8), 4)
        if kind == "k":
            acc = vr(Y + 16 * half, 16)
            c = acc if seen_first[half] else "0"
            seen_first[half] = True
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fr, ar(Q + 4 * step, 4), c))
        else:
            acc = ar(OACC + 16 * half, 16)
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, fr, vr(U + 4 * step, 4), acc))
        if j + AHEAD < n_m:
            lines.append(read(j + AHEAD))
        elif prefetch_next is not None and j + AHEAD - n_m < AHEAD:
            lines.append(first_reads(prefetch_next)[j + AHEAD - n_m])
        if dma_groups and j < len(dma_groups):
            lines += dma_groups[j]
        for k in range(per_gap):
            if gi < len(groups):
                if k > 0:
                    lines.append("s_nop 1")           # the next group reads what the exps just above wrote
                lines += groups[gi]
                gi += 1
    while gi < len(groups):
        lines.append("s_nop 1")
        lines += groups[gi]
        gi += 1
    return lines


def check_and_count(fail_label):
    return ["v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(TS), "s_nop 4", "s_cbranch_vccnz %s" % fail_label,   # !(2^80 > sum)
            "v_add_f32 %%[l], %%[l], %s" % v(TS), "s_add_u32 %[t], %[t], 1"]


def emit():
    L = []
    L += ["v_accvgpr_write_b32 a%d, %%[o%d]" % (i, i) for i in range(64)]
    L += ["global_load_dwordx4 %s, %%[qvo], %%[qb] offset:%d" % (ar(Q + 4 * ks, 4), 32 * ks) for ks in range(8)]
    roles = {1: (SA, SB, PA, PB), 2: (SB, SA, PB, PA), 3: (SA, SB, PA, PB), 0: (SB, SA, PB, PA)}
    # ---- warm-up at phase 1 ----
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    head, groups = top_protocol(1)
    L += head + [ln for g in groups for ln in g]
    X, Y, U, W = roles[1]
    L += iteration(0, Y, X, U, W, pv=False, softmax=False)         # QK(t) alone -> SA (K(t) sits in slot 1)
    L += ["s_nop 15", "s_nop 15"]
    L += iteration(1, X, Y, U, W, pv=False, prefetch_next=2)        # QK(t+1) -> SB under softmax(t) -> PB
    L += check_and_count("90f")
    L += ["s_branch 12f"]
    # ---- the loop: phases 1, 2, 3, 0 ----
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
    # ---- drain: PV of the last tile (its P is the W of phase 0) ----
    X, Y, U, W = roles[0]
    L += iteration(1, Y, X, W, U, pv=True, softmax=False, qk=False, reads_in_flight=True)
    L += ["s_mov_b32 %[code], 0", "s_branch 99f"]
    L += ["90:", "s_mov_b32 %[code], 1"]
    L += ["99:", "s_nop 15", "s_nop 15"]
    L += ["v_accvgpr_read_b32 %%[o%d], a%d" % (i, i) for i in range(64)]
    L += ["s_waitcnt lgkmcnt(0)"]
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("ATTN128_PIPE_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "attn128_pipe_loop.inc")
    lines = emit()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_attn128_pipe.py -- do not edit.  Steady-state KV loop of the pipelined d = 128 attention.\n")
        f.write("#define ALG_ATTN128_PIPE_LOOP_ASM \\\n")
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
        regs = ["a%d" % i for i in range(128)] + ["v%d" % i for i in range(64, 170)]
        f.write("#define ALG_ATTN128_PIPE_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
        f.write("#define ALG_ATTN128_PIPE_O_OPERANDS(o) \\\n  " + ", ".join('[o%d] "+v"(o[%d])' % (i, i) for i in range(64)) + '\n')
    print("wrote", os.path.normpath(path), len(lines), "lines,", sum(1 for l in lines if l.startswith("v_mfma")), "MFMAs")


if __name__ == "__main__":
    main()
