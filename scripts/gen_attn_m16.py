#!/usr/bin/env python3
"""Generates alg_amd/csrc/attn64_m16_loop.inc: the steady-state KV loop of the d = 64 attention on v_mfma_f32_16x16x32_bf16.

Round 6.  Same construction as gen_attn_pipe.py's 8-wave statement (one 256-query unit per workgroup, eight waves x 32 queries,
K / V^T tiles of 64 through four-slot LDS-DMA rings, PV(t-1) / QK(t+1) / softmax(t) in ONE asm statement, pre-scaled scores with a
zero offset, lazy running max: the statement leaves on a row sum >= 2^80) on the OTHER bf16 MFMA shape: under the 1400 W package
cap a register-only loop of 16x16x32 sustains ~10 % more than 32x32x16 (scripts/micro/mfma_shape.hip,
profiles/r6_mfma_shape_power.txt), and this kernel is power-bound.

Per iteration (64 keys) and wave: 32 MFMAs of 16x16x32 (= the 16 of 32x32x16), 16 fragment reads (each fragment feeds the wave's TWO
16-query blocks), one K and one V^T LDS-DMA piece, 32 v_exp + 16 v_cvt_pk + 32 v_add: per MFMA gap ONE exp, ONE add and every
other gap a pack -- none of it depends on this iteration's MFMAs.

Layouts (lane l: r = l & 15, g = l >> 4):
  S^T block (j, u) x query block qb (16 keys x 16 queries, 4 registers): lane holds query 16 qb + r and the four keys
      key(j, u, g, e) = 32 j + 16 (g >> 1) + 8 u + 4 (g & 1) + e                                   (e = register 0..3)
  i.e. the K fragment of block (j, u) takes the tile's key rows {0..7, 16..23} + 32 j + 8 u (row part r + 8 (r >> 3)).  With that
  choice the two S blocks (j, 0), (j, 1) of a lane ARE the B operand of the PV MFMA over keys [32 j, 32 j + 32) in the order V^T is
  stored in (kv index bits 2 <-> 3 swapped per 16, the layout the V^T projection writes for every d = 64 kernel): the lane's
  k-group g = positions 8 g .. 8 g + 7 = keys 16 (g >> 1) + {4 (g & 1) + e, 8 + 4 (g & 1) + e} -- P never leaves its registers
  and the global V^T layout is unchanged.
  K tile in the LDS: [64 rows][128 B], 16-byte chunk c of row y at c ^ swK(y), swK(y) = ((y & 7) >> 1) | ((y >> 4) & 1) << 2 -- it
  depends on the row PART only (not on j, u), so block (j, u) is an immediate offset, and every 16-lane read group covers all
  sixteen bank groups.  V^T tile: [64 d][128 B], chunk c of row y at c ^ ((y >> 1) & 7) (unchanged).
  O^T block (db, qb) = a[4 (2 db + qb) .. + 3]: lane holds query 16 qb + r, d = 16 db + 4 g + e.

Register plan (8-wave workgroup: hipcc grants 128 + 128 registers per lane; named literally, clobbered):
    v[26:57] SA  v[58:89] SB      score tiles: block (j, u), query block qb at + 4 (2 (2 j + u) + qb); roles alternate with t & 1
    v[90:105] PA v[106:121] PB    packed probabilities: operand (qb, j) = 4 registers at + 4 (2 qb + j): (u = 0: e01, e23, u = 1: e01, e23)
    v122 / v123 row sums of the tile (qb = 0 / 1), v[124:127] exp results in flight (v127 doubles as the check's scratch)
    a[0:31] O^T, a[32:63] eight fragment buffers (ring: fragment f in buffer f & 7, read four fragments = eight MFMAs ahead),
    a[64:79] Q fragments (qb, ks) at + 4 (2 qb + ks)
Operands: o0..o31 "+a", l0 / l1 "+v" running row sums (qb = 0 / 1), t "+s", code "=s" (as gen_attn_pipe.py), lk0 / lk1 "v" LDS byte
address of the lane's K fragment for k-step ks (d chunk 4 ks + g; ring base included), lv0 / lv1 "v" ... of its V^T fragment for key
half j, kvo0 / vvo0 "+v" DMA byte offsets of the lane into the K / V^T panels at tile (t + 3) / (t + 2) (advanced inside), qvo0 / qvo1
"v" byte offset of the lane's Q fragment row for qb = 0 / 1, kb / vb / qb "s" 64-bit panel bases, kstep "s" bytes per K tile, tend "s",
wk / wv "s" LDS byte address of the K / V^T ring + wave * 1024.
Collective protocol: gen_attn_pipe.py's (top of iteration t: s_waitcnt vmcnt(2); s_barrier; DMA K(t + 3), V^T(t + 2)).
"""
import os
import re

VBASE = 26
SA, SB, PA, PB = VBASE, VBASE + 32, VBASE + 64, VBASE + 80
TS0, TS1 = VBASE + 96, VBASE + 97
E = [VBASE + 98 + i for i in range(4)]
OACC, FR, Q = 0, 32, 64
TILE = 8192
AHEAD = 4          # fragments read ahead (each fragment = two MFMAs)

v = lambda i: "v%d" % i
vr = lambda i, n: "v[%d:%d]" % (i, i + n - 1)
ar = lambda i, n: "a[%d:%d]" % (i, i + n - 1)


def s_reg(S, j, u, qb, e=0):
    return S + 4 * (2 * (2 * j + u) + qb) + e


def p_reg(P, qb, j, k=0):
    return P + 4 * (2 * qb + j) + k


# the 16 fragments of a full iteration, in MFMA order: ("v", db, j) = V^T(t-1) d-block db, key half j -> PV; ("k", j, u, ks) -> QK.
# The first four do not depend on the iteration's barrier (V^T(t-1) landed two iterations earlier); every S block is complete at
# least four MFMAs before the iteration ends (the next iteration's first exps read it).
FRAGS_FULL = [("v", 0, 0), ("v", 1, 0), ("v", 2, 0), ("v", 3, 0),
              ("k", 0, 0, 0), ("k", 0, 0, 1), ("k", 0, 1, 0), ("k", 0, 1, 1), ("v", 0, 1),
              ("k", 1, 0, 0), ("k", 1, 0, 1), ("v", 1, 1), ("k", 1, 1, 0), ("k", 1, 1, 1), ("v", 2, 1), ("v", 3, 1)]
FRAGS_QK = [f for f in FRAGS_FULL if f[0] == "k"]
FRAGS_PV = [f for f in FRAGS_FULL if f[0] == "v"]

# the order softmax(t) walks the 32 scores of a lane: block (j, u) outer (the order QK completes them), query block, register
SCORES = [(j, u, qb, e) for j in range(2) for u in range(2) for qb in range(2) for e in range(4)]


def frag_read(buf, f, kslot, vslot):
    if f[0] == "k":
        _, j, u, ks = f
        return "ds_read_b128 %s, %%[lk%d] offset:%d" % (ar(FR + 4 * buf, 4), ks, kslot * TILE + (32 * j + 8 * u) * 128)
    _, db, j = f
    return "ds_read_b128 %s, %%[lv%d] offset:%d" % (ar(FR + 4 * buf, 4), j, vslot * TILE + db * 2048)


def mfmas(f, buf, Y, U, first_k):
    """the two MFMAs (query blocks 0, 1) of fragment f sitting in ring buffer buf"""
    fr = ar(FR + 4 * buf, 4)
    out = []
    for qb in range(2):
        if f[0] == "k":
            _, j, u, ks = f
            acc = vr(s_reg(Y, j, u, qb), 4)
            c = "0" if ks == 0 else acc
            out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc, fr, ar(Q + 4 * (2 * qb + ks), 4), c))
        else:
            _, db, j = f
            acc = ar(OACC + 4 * (2 * db + qb), 4)
            out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc, fr, vr(p_reg(U, qb, j), 4), acc))
    return out


def softmax_groups(X, W, pre):
    """VALU groups of softmax(t): group n = [pack of pair n / 2 (n even)], add of score n, exp of score n + pre.  pre = 2: the
    caller issued the exps of scores 0, 1 (softmax_pre); pre = 0 (warm-up): the groups start two early with bare exps."""
    groups = []
    n_lo = -pre_shift(pre)
    for n in range(n_lo, 32):
        g = []
        if n >= 0 and n % 2 == 0:
            j, u, qb, e = SCORES[n]
            g.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(p_reg(W, qb, j, 2 * u + (e >> 1))), v(E[n & 3]), v(E[(n + 1) & 3])))
        if n >= 0:
            j, u, qb, e = SCORES[n]
            ts = TS0 if qb == 0 else TS1
            first = (u == 0 and j == 0 and e == 0)
            # the tile's first score of a query block starts its sum
            g.append(("v_mov_b32 %s, %s" % (v(ts), v(E[n & 3]))) if first else ("v_add_f32 %s, %s, %s" % (v(ts), v(ts), v(E[n & 3]))))
        if n + 2 < 32:
            j, u, qb, e = SCORES[n + 2]
            g.append("v_exp_f32 %s, %s" % (v(E[(n + 2) & 3]), v(s_reg(X, j, u, qb, e))))
        groups.append(g)
    return groups


def pre_shift(pre):
    return 2 - pre


def softmax_pre(X):
    """the exps of scores 0 and 1, issued AHEAD of the top-of-iteration wait + barrier (the scores are complete)"""
    out = []
    for n in range(2):
        j, u, qb, e = SCORES[n]
        out.append("v_exp_f32 %s, %s" % (v(E[n & 3]), v(s_reg(X, j, u, qb, e))))
    return out


def top_protocol(phase, spread=False):
    """top of iteration t (t & 3 == phase): all but the previous iteration's two DMAs have landed, everybody is done with iteration
    t - 1; then this wave's piece of K(t + 3) and of V^T(t + 2) (eight waves stage an 8 KiB tile)"""
    ks, vs = (phase + 3) & 3, (phase + 2) & 3
    head = ["s_waitcnt vmcnt(2)", "s_barrier"]
    groups = [["s_add_u32 m0, %%[wk], %d" % (ks * TILE), "s_nop 0", "global_load_lds_dwordx4 %[kvo0], %[kb]",
               "v_add_u32 %[kvo0], %[kstep], %[kvo0]"],
              ["s_add_u32 m0, %%[wv], %d" % (vs * TILE), "s_nop 0", "global_load_lds_dwordx4 %[vvo0], %[vb]",
               "v_add_u32 %[vvo0], 0x80, %[vvo0]"]]
    if spread:
        return head, groups
    return head + [ln for g in groups for ln in g]


def iteration(phase, X, Y, U, W, pv=True, softmax=True, qk=True, reads_in_flight=False, prefetch_next=None, dma_groups=None, pre=0):
    """one pipelined iteration at ring phase t & 3 == phase: fragments in FRAGS order, two MFMAs each; fragment f + AHEAD is read
    right behind the first MFMA of fragment f; the softmax groups are spread evenly over the MFMA gaps"""
    kslot, vslot = (phase + 1) & 3, (phase - 1) & 3
    frags = FRAGS_FULL if (pv and qk) else (FRAGS_QK if qk else FRAGS_PV)
    nf = len(frags)
    lines = []
    if not reads_in_flight:
        for f in range(min(AHEAD, nf)):
            lines.append(frag_read(f & 7, frags[f], kslot, vslot))
    groups = softmax_groups(X, W, pre) if softmax else []
    n_gaps = 2 * nf
    per_gap = -(-len(groups) // n_gaps) if groups else 0
    gi = 0
    for f, fr in enumerate(frags):
        two = mfmas(fr, f & 7, Y, U, None)
        for h in range(2):
            gap = 2 * f + h
            lines.append(two[h])
            if h == 0:
                if f + AHEAD < nf:
                    lines.append(frag_read((f + AHEAD) & 7, frags[f + AHEAD], kslot, vslot))
                elif prefetch_next is not None:
                    # the next iteration's first fragments (V^T of the tile this iteration's softmax is about: barrier-free)
                    nxt_v = (prefetch_next - 1) & 3
                    lines.append(frag_read((f + AHEAD - nf) & 7, FRAGS_FULL[f + AHEAD - nf], None, nxt_v))
            if dma_groups and gap < len(dma_groups):
                lines += dma_groups[gap]
            for k in range(per_gap):
                if gi < len(groups):
                    if k > 0:
                        lines.append("s_nop 1")     # a pack right behind the exp it reads (trans -> VALU use)
                    lines += groups[gi]
                    gi += 1
    while gi < len(groups):
        lines.append("s_nop 1")
        lines += groups[gi]
        gi += 1
    return lines


def check_and_count(fail_label):
    """row-sum check of the iteration just issued (both query blocks at once: their sum), then t += 1"""
    return ["v_add_f32 %s, %s, %s" % (v(E[3]), v(TS0), v(TS1)), "v_cmp_ngt_f32 vcc, 0x67800000, %s" % v(E[3]), "s_nop 4",
            "s_cbranch_vccnz %s" % fail_label,   # !(2^80 > sum): also inf and NaN
            "v_add_f32 %%[l0], %%[l0], %s" % v(TS0), "v_add_f32 %%[l1], %%[l1], %s" % v(TS1), "s_add_u32 %[t], %[t], 1"]


_DS = re.compile(r"^ds_read_b128 a\[(\d+):\d+\]")
_MF = re.compile(r"^v_mfma_f32_16x16x32_bf16 \S+ a\[(\d+):\d+\],")


def place_lgkm_waits(lines, entry_labels):
    """the loosest correct `s_waitcnt lgkmcnt(n)` in front of every MFMA whose fragment is not known to have arrived (the LDS
    returns in order); the queue of reads in flight must be the same wherever a label in entry_labels is entered from"""
    out, queue, at_label = [], [], {}

    def check(label, q):
        if label in at_label:
            assert at_label[label] == q, ("fragment reads in flight differ at label %s" % label, at_label[label], q)
        else:
            at_label[label] = list(q)
    for i, ln in enumerate(lines):
        m = re.match(r"^(\d+):$", ln)
        if m and m.group(1) in entry_labels:
            check(m.group(1), queue)
        b = re.match(r"^s_cbranch_\w+ (\d+)[bf]$", ln) or re.match(r"^s_branch (\d+)[bf]$", ln)
        if b and b.group(1) in entry_labels:
            check(b.group(1), queue)
        if ln.startswith("s_branch"):
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
        if f and int(f.group(1)) in queue:
            idx = max(k for k, reg in enumerate(queue) if reg == int(f.group(1)))
            out.append("s_waitcnt lgkmcnt(%d)" % (len(queue) - 1 - idx))
            queue = queue[idx + 1:]
        out.append(ln)
    return out


def emit():
    L = []
    # ---- entry: O -> a[0:31], Q fragments ----
    L += ["v_accvgpr_mov_b32 a%d, %%[o%d]" % (i, i) for i in range(32)]
    L += ["global_load_dwordx4 %s, %%[qvo%d], %%[qb] offset:%d" % (ar(Q + 4 * (2 * qb + ks), 4), qb, 64 * ks)
          for qb in range(2) for ks in range(2)]
    roles = {1: (SA, SB, PA, PB), 2: (SB, SA, PB, PA), 3: (SA, SB, PA, PB), 0: (SB, SA, PB, PA)}
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]   # Q (and, once, whatever the caller had in flight)
    # ---- warm-up at phase 1: top protocol, QK(t) alone into X = SA (K(t) sits in slot 1 = the "next" slot of phase 0) ----
    L += top_protocol(1)
    X, Y, U, W = roles[1]
    L += iteration(0, Y, X, U, W, pv=False, softmax=False)        # S(t) -> SA
    L += ["s_nop 15", "s_nop 15"]                                    # S(t) complete before the first exp reads it
    L += iteration(1, X, Y, U, W, pv=False, prefetch_next=2)        # QK(t+1) -> SB under softmax(t) -> PB
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
        L += iteration(ph, X, Y, U, W, reads_in_flight=True, prefetch_next=(ph + 1) & 3, dma_groups=groups, pre=2)
        L += check_and_count("90f")
    L += ["s_add_u32 %[code], %[t], 4", "s_cmp_le_u32 %[code], %[tend]", "s_cbranch_scc1 11b"]
    # ---- drain: PV of the last tile (its P sits in W of phase 0 = PA) ----
    X, Y, U, W = roles[0]
    L += iteration(1, Y, X, W, U, pv=True, softmax=False, qk=False, reads_in_flight=True)
    L += ["s_mov_b32 %[code], 0", "s_branch 99f"]
    # ---- failed row-sum check in iteration t: its MFMAs are issued; leave with code 1 ----
    L += ["90:", "s_mov_b32 %[code], 1"]
    L += ["99:", "s_nop 15", "s_nop 15"]
    L += ["v_accvgpr_mov_b32 %%[o%d], a%d" % (i, i) for i in range(32)]
    L += ["s_waitcnt lgkmcnt(0)"]
    return place_lgkm_waits(L, {"11", "12"})


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("ATTN_M16_OUT") or os.path.join(here, "..", "alg_amd", "csrc", "attn64_m16_loop.inc")
    lines = emit()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_attn_m16.py -- do not edit.  Steady-state KV loop of the d = 64 attention on v_mfma_f32_16x16x32_bf16.\n")
        f.write("#define ALG_ATTN_M16_LOOP_ASM \\\n")
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
        regs = ["a%d" % i for i in range(80)] + ["v%d" % i for i in range(VBASE, VBASE + 102)]
        f.write("#define ALG_ATTN_M16_CLOBBERS \\\n  " + ", ".join('"%s"' % r for r in regs) + '\n')
        f.write("#define ALG_ATTN_M16_O_OPERANDS(o) \\\n  " + ", ".join('[o%d] "+a"(o[%d])' % (i, i) for i in range(32)) + '\n')
    print("wrote", os.path.normpath(path), len(lines), "lines,", sum(1 for l in lines if l.startswith("v_mfma")), "MFMAs")


if __name__ == "__main__":
    main()
