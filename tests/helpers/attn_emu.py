"""Harness that runs a generated 64-queries-per-wave attention statement (scripts/gen_attn_q64.py) in the instruction-level
emulator (scripts/asm_emu.py) exactly the way the C++ frame (alg_amd/csrc/attention128_q64.hip) drives it: the same per-lane
operand values, the same LDS ring layout and DMA lane mapping, tile 0 and the tail tiles done in numpy in the frame's place."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import asm_emu  # noqa: E402
import gen_attn_q64 as G  # noqa: E402

KVB = 64


def perm8(kv):
    """physical V^T column of key kv: index bits 2 and 3 swapped (the GEMM that writes V^T stores it so)"""
    return (kv & ~12) | ((kv & 4) << 1) | ((kv & 8) >> 1)


def bf16_round(x):
    return asm_emu.bf16_to_f32(asm_emu.f32_to_bf16(np.asarray(x, dtype=np.float32)))


class Problem:
    def __init__(self, d, T, seed=0, q_rs=None, k_rs=None, score_scale=1.0, prescaled=False, ragged=0):
        """prescaled: Q carries scale * log2(e) already (the d = 64 frame: alg_qk_norm_rope_scaled), scores arrive in log2 units
        and the statement runs with a ZERO offset (no subtraction at all).  ragged: keys missing from the last of the T tiles."""
        rng = np.random.default_rng(seed)
        self.d, self.T, self.Skv, self.Sq = d, T, T * KVB - ragged, 256
        self.ragged = ragged
        self.q_rs, self.k_rs = q_rs or 2 * d, k_rs or 2 * d          # row pitches in elements (a [S, 2, d] qk tensor by default)
        self.vt_rs = self.T * KVB + 64
        c_true = 1.0 / np.sqrt(d) * 1.4426950408889634
        self.prescaled = prescaled
        self.q = bf16_round(rng.standard_normal((self.Sq, d)) * score_scale * (c_true if prescaled else 1.0))
        self.k = bf16_round(rng.standard_normal((self.Skv, d)))
        self.v = bf16_round(rng.standard_normal((self.Skv, d)))
        self.c = np.float32(1.0 if prescaled else c_true)           # what one raw score unit is worth in log2 units
        # global memory: Q panel, K panel, V^T panel (permuted columns), each at an odd offset to catch base mix-ups
        self.QOFF, self.KOFF = 4096, 4096 + 2 * self.Sq * self.q_rs + 512
        self.VOFF = self.KOFF + 2 * self.Skv * self.k_rs + 1024        # NOTHING valid behind the last K row: a DMA must not go there
        self.pack()

    def pack(self):
        """(re)build the global-memory image from q / k / v"""
        d = self.d
        size = self.VOFF + 2 * d * self.vt_rs + 256
        g16 = np.full(size // 2, 0x7FC0, dtype=np.uint16)              # bf16 NaN everywhere: data fetched from outside a panel poisons the result
        for dd in range(d):                                            # V^T pad columns (up to the pitch) are ZERO by contract
            g16[self.VOFF // 2 + dd * self.vt_rs: self.VOFF // 2 + (dd + 1) * self.vt_rs] = 0
        tobf = lambda a: asm_emu.f32_to_bf16(a)
        for r in range(self.Sq):
            g16[self.QOFF // 2 + r * self.q_rs: self.QOFF // 2 + r * self.q_rs + d] = tobf(self.q[r])
        for r in range(self.Skv):
            g16[self.KOFF // 2 + r * self.k_rs: self.KOFF // 2 + r * self.k_rs + d] = tobf(self.k[r])
        cols = perm8(np.arange(self.Skv))
        for dd in range(d):
            g16[self.VOFF // 2 + dd * self.vt_rs + cols] = tobf(self.v[:, dd])
        self.gmem = g16.view(np.uint8)

    def reference(self):
        s = self.q.astype(np.float64) @ self.k.astype(np.float64).T * float(self.c)      # log2 units
        p = np.exp2(s - s.max(axis=1, keepdims=True))
        return (p / p.sum(axis=1, keepdims=True)) @ self.v.astype(np.float64)


def k_dma_lane(d, tid, i):
    """(K row within the tile, LOGICAL 16-byte slot of that row) thread `tid` fetches in DMA piece i -- the frames' stage_k:
    d = 128: 256-byte rows, 16 slots, 16 rows per piece, physical slot p holds logical p ^ (row & 15);
    d = 64:  128-byte rows, 8 slots, 32 rows per piece, physical slot p holds logical p ^ ((row >> 1) & 7)"""
    if d == 128:
        row = (tid >> 4) + 16 * i
        return row, (tid & 15) ^ (row & 15)
    row = (tid >> 3) + 32 * i
    return row, (tid & 7) ^ ((row >> 1) & 7)


def k_frag_addr(d, l31, h2, ks):
    """byte offset (inside a K tile, sub-tile 0) of the fragment lane (l31, h2) reads for k-step ks"""
    if d == 128:
        return l31 * 256 + (((2 * ks + h2) ^ (l31 & 15)) * 16)
    return l31 * 128 + (((2 * ks + h2) ^ ((l31 >> 1) & 7)) * 16)


def lane_ctx(wave, lane):
    l31, h2, tid = lane & 31, lane >> 5, wave * 64 + lane
    return l31, h2, tid


def stage(pb, lds, cfg, which, tile, slot_base):
    """what the four waves' DMA pieces of one K / V^T tile put into ring slot tile & 3 (frame: stage_k / stage_v)"""
    d = cfg.d
    for wave in range(4):
        for lane in range(64):
            l31, h2, tid = lane_ctx(wave, lane)
            for i in range(cfg.NP // 2):
                if which == "k":
                    row, slot = k_dma_lane(d, tid, i)
                    src = pb.KOFF + (min(tile * KVB + row, pb.Skv - 1) * pb.k_rs + slot * 8) * 2
                else:
                    row = tid // 8 + 32 * i
                    slot = (tid & 7) ^ ((tid >> 4) & 7)
                    src = pb.VOFF + (row * pb.vt_rs + slot * 8 + tile * KVB) * 2
                dst = slot_base + (tile & 3) * cfg.TILE + (i * 4 + wave) * 1024 + lane * 16
                lds[dst:dst + 16] = pb.gmem[src:src + 16]


def run_statement(pb, cfg, lazy_reads, lazy_dma, t0=1, tend=None, mutate=None, va=None):
    """Emulates the statement entered at iteration t0 (tile 0 .. t0 - 1 done by `the frame` = numpy here).  Returns the
    normalised attention output [Sq, d] after the frame's tail, the exit iteration and the exit code of wave 0."""
    d, T, c = cfg.d, pb.T, pb.c
    tend = (T - 2 if pb.ragged else T - 1) if tend is None else tend     # iterations t < tend run: QK(t + 1) stays unmasked
    lines = G.emit(cfg)
    if mutate is not None:
        lines = mutate(lines)
    KL, VL = 0, 4 * cfg.TILE
    # ---- operand binding: "v" operands in v0 .., "s" in s0 .., O tuples in a[0 : 16 NO) ----
    tab, nv, ns = {}, 0, 0
    def vreg(name):
        nonlocal nv
        tab[name] = "v%d" % nv
        nv += 1
    def sreg(name, n=1):
        nonlocal ns
        ns = (ns + n - 1) // n * n
        tab[name] = "s%d" % ns if n == 1 else "s[%d:%d]" % (ns, ns + n - 1)
        ns += n
    for i in range(cfg.NO):
        tab["o%d" % i] = "a[%d:%d]" % (16 * i, 16 * i + 15)
    for n in ["l0", "l1"] + (["negmc0", "negmc1"] if cfg.fma else []) + ["qvo0", "qvo1"] + ["lk%d" % i for i in range(cfg.KS)] + ["lv%d" % i for i in range(4)] + \
             ["kvo%d" % i for i in range(cfg.NP // 2)] + ["vvo%d" % i for i in range(cfg.NP // 2)]:
        vreg(n)
    assert nv <= cfg.VB
    for n in ("t", "code") + (("c",) if cfg.fma else ()) + ("kstep", "tend", "wk", "wv") + tuple("kd%d" % i for i in range(4)) + \
             tuple("vd%d" % i for i in range(4)):
        sreg(n)
    sreg("qb", 2)
    assert ns <= G.KD
    # the memory image sits at virtual address `va`; by default the K panel crosses a 4 GiB boundary between its tiles 6 and 7, so
    # the carry of the descriptor's base advance is exercised
    va = ((1 << 32) - pb.KOFF - 6 * KVB * pb.k_rs * 2 - 64) if va is None else va
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=4, gmem=pb.gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma, gmem_va=va)
    # ---- the frame's state at entry: tiles 0 .. t0 - 1 folded into (m, l, O); K(t0 - 1 .. t0 + 2), V(t0 - 1 .. t0 + 1) resident ----
    qf, kf, vf = pb.q.astype(np.float64), pb.k.astype(np.float64), pb.v.astype(np.float64)
    s_all = qf @ kf.T                                       # raw scores [Sq, Skv]
    m_run = s_all[:, :KVB].max(axis=1)                      # the frame takes tile 0's maximum as the (lazy) offset
    if not cfg.fma:
        m_run = np.zeros(pb.Sq)                             # pre-scaled zero-offset form: p = exp2(s)
    def probs(t):
        return bf16_round(np.exp2((s_all[:, t * KVB:(t + 1) * KVB] - m_run[:, None]) * float(c))).astype(np.float64)
    def fsum(t):
        return np.exp2((s_all[:, t * KVB:(t + 1) * KVB] - m_run[:, None]) * float(c)).astype(np.float32).astype(np.float64).sum(axis=1)
    O = np.zeros((pb.Sq, d))
    l = np.zeros(pb.Sq)
    for t in range(t0):
        O += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l += fsum(t)
    for t in range(t0 - 1, t0 + 3):
        stage(pb, m.lds, cfg, "k", t, KL)
    for t in range(t0 - 1, t0 + 2):
        stage(pb, m.lds, cfg, "v", t, VL)
    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        if r[2] == 1:
            w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        else:
            w.s[r[1]], w.s[r[1] + 1] = np.uint32(int(val) & 0xFFFFFFFF), np.uint32(int(val) >> 32)
    def vset(w, name, arr):
        w.v[asm_emu.parse_reg(tab[name])[1]] = np.asarray(arr).astype(np.int64).astype(np.uint32) if np.asarray(arr).dtype != np.float32 else np.asarray(arr).view(np.uint32)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2, tid = lane & 31, lane >> 5, w.id * 64 + lane
        q_row = w.id * 64 + l31
        sset(w, "t", t0), sset(w, "tend", tend), sset(w, "kstep", KVB * pb.k_rs * 2)
        if cfg.fma:
            sset(w, "c", int(np.float32(c).view(np.uint32)))
        sset(w, "wk", KL + w.id * 1024), sset(w, "wv", VL + w.id * 1024)
        sset(w, "qb", va + pb.QOFF)
        # the frame's descriptors at entry: base = the tile the first DMA fetches (K(t0 + 3), V^T(t0 + 2)), stride 0, num_records =
        # what is left of the panel from there (K: up to the end of its last row's head slice; V^T: d rows of the pitch)
        kbytes, vbytes = ((pb.Skv - 1) * pb.k_rs + d) * 2, d * pb.vt_rs * 2
        koff, voff = (t0 + 3) * KVB * pb.k_rs * 2, (t0 + 2) * KVB * 2
        for i, val in enumerate(((va + pb.KOFF + koff) & 0xFFFFFFFF, (va + pb.KOFF + koff) >> 32, max(kbytes - koff, 0), 0x00020000)):
            sset(w, "kd%d" % i, val)
        for i, val in enumerate(((va + pb.VOFF + voff) & 0xFFFFFFFF, (va + pb.VOFF + voff) >> 32, max(vbytes - voff, 0), 0x00020000)):
            sset(w, "vd%d" % i, val)
        for qh in range(2):
            vset(w, "qvo%d" % qh, ((q_row + 32 * qh) * pb.q_rs + h2 * 8) * 2)
            if cfg.fma:
                vset(w, "negmc%d" % qh, (-(m_run[q_row + 32 * qh]) * float(c)).astype(np.float32))
            vset(w, "l%d" % qh, np.where(h2 == 0, l[q_row + 32 * qh], 0.0).astype(np.float32))   # tile sums so far: all in lane h2 = 0
        for ks in range(cfg.KS):
            vset(w, "lk%d" % ks, KL + k_frag_addr(d, l31, h2, ks))
        for kk in range(4):
            vset(w, "lv%d" % kk, VL + l31 * 128 + (((2 * kk + h2) ^ ((l31 >> 1) & 7)) * 16))
        for i in range(cfg.NP // 2):
            row, slot = k_dma_lane(d, tid, i)
            vset(w, "kvo%d" % i, (row * pb.k_rs + slot * 8) * 2)                  # the lane's place inside tile 0
            vrow, vslot = tid // 8 + 32 * i, (tid & 7) ^ ((tid >> 4) & 7)
            vset(w, "vvo%d" % i, (vrow * pb.vt_rs + vslot * 8) * 2)
        for qh in range(2):
            for dt in range(cfg.DT):
                base = 16 * (qh * cfg.DT + dt)
                for e in range(16):
                    drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
                    w.a[base + e] = O[q_row + 32 * qh, drow].astype(np.float32).view(np.uint32)
    n = m.run()
    # ---- read back, then the frame's tail in numpy ----
    t_exit = int(m.waves[0].s[asm_emu.parse_reg(tab["t"])[1]])
    codes = [int(w.s[asm_emu.parse_reg(tab["code"])[1]]) for w in m.waves]
    O2, l2 = np.zeros((pb.Sq, d)), np.zeros(pb.Sq)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2 = lane & 31, lane >> 5
        q_row = w.id * 64 + l31
        for qh in range(2):
            lv = w.v[asm_emu.parse_reg(tab["l%d" % qh])[1]].view(np.float32).astype(np.float64)
            np.add.at(l2, q_row + 32 * qh, lv)
            for dt in range(cfg.DT):
                base = 16 * (qh * cfg.DT + dt)
                for e in range(16):
                    drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
                    O2[q_row + 32 * qh, drow] = w.a[base + e].view(np.float32)
    for t in range(t_exit, T):
        O2 += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l2 += fsum(t)
    return O2 / l2[:, None], t_exit, codes, n, lines


# ---------------------------------------------------------------------------------------------------------------------
# the DEFAULT d = 64 kernel's statement (scripts/gen_attn_pipe.py, 8-wave form: 32 queries per wave, attention.hip's frame)
# ---------------------------------------------------------------------------------------------------------------------
def run_pipe8_statement(pb, lazy_reads, lazy_dma, t0=1, mutate=None):
    """flash_attn_d64_pipe_kernel<8>'s statement for one 256-query unit: eight waves x 32 queries, K / V^T tiles of 8 KiB through
    four-slot rings (one 1 KiB piece of each per wave and tile), O in 32 scalar "+a" operands, pre-scaled scores with a zero
    offset, whole groups of four iterations while t + 4 <= tend = T - 3 (this statement's own, older exit rule)."""
    import gen_attn_pipe as GP
    assert pb.d == 64 and pb.prescaled and not pb.ragged
    GP.configure(8)
    lines = GP.emit()
    GP.configure(4)
    if mutate is not None:
        lines = mutate(lines)
    T, TILE, NW = pb.T, 8192, 8
    tend = T - 3
    KL, VL = 0, 4 * TILE
    tab = {"o%d" % i: "a%d" % (80 + i) for i in range(32)}
    names_v = ["l", "kvo0", "vvo0", "qvo"] + ["lk%d" % i for i in range(4)] + ["lv%d" % i for i in range(4)]
    for i, n in enumerate(names_v):
        tab[n] = "v%d" % i
    assert len(names_v) <= 26
    for i, n in enumerate(("t", "code", "kstep", "tend", "wk", "wv")):
        tab[n] = "s%d" % i
    for i, n in enumerate(("kb", "vb", "qb")):
        tab[n] = "s[%d:%d]" % (8 + 2 * i, 9 + 2 * i)
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=NW, gmem=pb.gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma)
    qf, kf, vf = pb.q.astype(np.float64), pb.k.astype(np.float64), pb.v.astype(np.float64)
    s_all = qf @ kf.T                                      # log2 units already
    probs = lambda t: bf16_round(np.exp2(s_all[:, t * KVB:(t + 1) * KVB])).astype(np.float64)
    fsum = lambda t: np.exp2(s_all[:, t * KVB:(t + 1) * KVB]).astype(np.float32).astype(np.float64).sum(axis=1)
    O, l = np.zeros((pb.Sq, 64)), np.zeros(pb.Sq)
    for t in range(t0):
        O += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l += fsum(t)

    def stage8(which, tile):
        for wave in range(NW):
            lane = np.arange(64)
            tid = wave * 64 + lane
            row, slot = tid >> 3, (tid & 7) ^ ((tid >> 4) & 7)
            for ln in range(64):
                if which == "k":
                    src = pb.KOFF + ((tile * KVB + row[ln]) * pb.k_rs + slot[ln] * 8) * 2
                    dst = KL + (tile & 3) * TILE + wave * 1024 + ln * 16
                else:
                    src = pb.VOFF + (row[ln] * pb.vt_rs + slot[ln] * 8 + tile * KVB) * 2
                    dst = VL + (tile & 3) * TILE + wave * 1024 + ln * 16
                m.lds[dst:dst + 16] = pb.gmem[src:src + 16]
    for t in range(t0 - 1, t0 + 3):
        stage8("k", t)
    for t in range(t0 - 1, t0 + 2):
        stage8("v", t)

    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        if r[2] == 2:
            w.s[r[1] + 1] = np.uint32(int(val) >> 32)

    def vset(w, name, arr):
        a = np.asarray(arr)
        w.v[asm_emu.parse_reg(tab[name])[1]] = a.view(np.uint32) if a.dtype == np.float32 else a.astype(np.int64).astype(np.uint32)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2, tid = lane & 31, lane >> 5, w.id * 64 + lane
        q_row = w.id * 32 + l31
        srow, sslot = tid >> 3, (tid & 7) ^ ((tid >> 4) & 7)
        sset(w, "t", t0), sset(w, "tend", tend), sset(w, "kstep", KVB * pb.k_rs * 2)
        sset(w, "wk", KL + w.id * 1024), sset(w, "wv", VL + w.id * 1024)
        sset(w, "kb", pb.KOFF), sset(w, "vb", pb.VOFF), sset(w, "qb", pb.QOFF)
        vset(w, "qvo", (q_row * pb.q_rs + h2 * 8) * 2)
        vset(w, "l", np.where(h2 == 0, l[q_row], 0.0).astype(np.float32))
        for ks in range(4):
            fl = l31 * 128 + (((2 * ks + h2) ^ ((l31 >> 1) & 7)) * 16)
            vset(w, "lk%d" % ks, KL + fl)
            vset(w, "lv%d" % ks, VL + fl)
        vset(w, "kvo0", (((t0 + 3) * KVB + srow) * pb.k_rs + sslot * 8) * 2)
        vset(w, "vvo0", (srow * pb.vt_rs + sslot * 8 + (t0 + 2) * KVB) * 2)
        for i in range(32):
            dt, e = i >> 4, i & 15
            drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
            w.a[80 + i] = O[q_row, drow].astype(np.float32).view(np.uint32)
    n = m.run()
    t_exit = int(m.waves[0].s[asm_emu.parse_reg(tab["t"])[1]])
    codes = [int(w.s[asm_emu.parse_reg(tab["code"])[1]]) for w in m.waves]
    O2, l2 = np.zeros((pb.Sq, 64)), np.zeros(pb.Sq)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2 = lane & 31, lane >> 5
        q_row = w.id * 32 + l31
        np.add.at(l2, q_row, w.v[asm_emu.parse_reg(tab["l"])[1]].view(np.float32).astype(np.float64))
        for i in range(32):
            dt, e = i >> 4, i & 15
            drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
            O2[q_row, drow] = w.a[80 + i].view(np.float32)
    for t in range(t_exit, T):
        O2 += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l2 += fsum(t)
    return O2 / l2[:, None], t_exit, codes, n, lines


def run_pipe128_statement(pb, lazy_reads, lazy_dma, t0=1, mutate=None):
    """flash_attn_d128_pipe_kernel's statement (scripts/gen_attn128_pipe.py; the 32-query d = 128 kernel: ALG_ATTN128_Q64=0 and
    sequences below the 64-query kernel's policy): four waves x 32 queries = the first 128 queries of the problem, O in 64 "+v"
    operands, the lazy offset as s * c - m * c, whole groups of four iterations while t + 4 <= tend = T - 3."""
    import gen_attn128_pipe as GP
    assert pb.d == 128 and not pb.prescaled and not pb.ragged
    lines = GP.emit()
    if mutate is not None:
        lines = mutate(lines)
    T, TILE, NW, c = pb.T, 16384, 4, float(pb.c)
    tend = T - 3
    KL, VL = 0, 4 * TILE
    tab = {"o%d" % i: "v%d" % i for i in range(64)}
    names_v = ["l", "negmc", "qvo"] + ["lk%d" % i for i in range(8)] + ["lv%d" % i for i in range(4)] + \
              ["kvo%d" % i for i in range(4)] + ["vvo%d" % i for i in range(4)]
    for i, n in enumerate(names_v):
        tab[n] = "v%d" % (170 + i)
    for i, n in enumerate(("t", "code", "c", "kstep", "tend", "wk", "wv")):
        tab[n] = "s%d" % i
    for i, n in enumerate(("kb", "vb", "qb")):
        tab[n] = "s[%d:%d]" % (8 + 2 * i, 9 + 2 * i)
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=NW, gmem=pb.gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma)
    nq = NW * 32
    qf, kf, vf = pb.q[:nq].astype(np.float64), pb.k.astype(np.float64), pb.v.astype(np.float64)
    s_all = qf @ kf.T
    m_run = s_all[:, :KVB].max(axis=1)
    probs = lambda t: bf16_round(np.exp2((s_all[:, t * KVB:(t + 1) * KVB] - m_run[:, None]) * c)).astype(np.float64)
    fsum = lambda t: np.exp2((s_all[:, t * KVB:(t + 1) * KVB] - m_run[:, None]) * c).astype(np.float32).astype(np.float64).sum(axis=1)
    O, l = np.zeros((nq, 128)), np.zeros(nq)
    for t in range(t0):
        O += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l += fsum(t)
    cfg = G.Cfg(128)                                      # (only for the staging helper: same tile geometry and lane mapping)
    for t in range(t0 - 1, t0 + 3):
        stage(pb, m.lds, cfg, "k", t, KL)
    for t in range(t0 - 1, t0 + 2):
        stage(pb, m.lds, cfg, "v", t, VL)

    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        if r[2] == 2:
            w.s[r[1] + 1] = np.uint32(int(val) >> 32)

    def vset(w, name, arr):
        a = np.asarray(arr)
        w.v[asm_emu.parse_reg(tab[name])[1]] = a.view(np.uint32) if a.dtype == np.float32 else a.astype(np.int64).astype(np.uint32)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2, tid = lane & 31, lane >> 5, w.id * 64 + lane
        q_row = w.id * 32 + l31
        sset(w, "t", t0), sset(w, "tend", tend), sset(w, "kstep", KVB * pb.k_rs * 2)
        sset(w, "c", int(np.float32(c).view(np.uint32)))
        sset(w, "wk", KL + w.id * 1024), sset(w, "wv", VL + w.id * 1024)
        sset(w, "kb", pb.KOFF), sset(w, "vb", pb.VOFF), sset(w, "qb", pb.QOFF)
        vset(w, "qvo", (q_row * pb.q_rs + h2 * 8) * 2)
        vset(w, "negmc", (-m_run[q_row] * c).astype(np.float32))
        vset(w, "l", np.where(h2 == 0, l[q_row], 0.0).astype(np.float32))
        for ks in range(8):
            vset(w, "lk%d" % ks, KL + k_frag_addr(128, l31, h2, ks))
        for kk in range(4):
            vset(w, "lv%d" % kk, VL + l31 * 128 + (((2 * kk + h2) ^ ((l31 >> 1) & 7)) * 16))
        for i in range(4):
            row, slot = k_dma_lane(128, tid, i)
            vset(w, "kvo%d" % i, (((t0 + 3) * KVB + row) * pb.k_rs + slot * 8) * 2)
            vrow, vslot = tid // 8 + 32 * i, (tid & 7) ^ ((tid >> 4) & 7)
            vset(w, "vvo%d" % i, (vrow * pb.vt_rs + vslot * 8 + (t0 + 2) * KVB) * 2)
        for i in range(64):
            dt, e = i >> 4, i & 15
            drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
            w.v[i] = O[q_row, drow].astype(np.float32).view(np.uint32)
    n = m.run()
    t_exit = int(m.waves[0].s[asm_emu.parse_reg(tab["t"])[1]])
    codes = [int(w.s[asm_emu.parse_reg(tab["code"])[1]]) for w in m.waves]
    O2, l2 = np.zeros((nq, 128)), np.zeros(nq)
    for w in m.waves:
        lane = np.arange(64)
        l31, h2 = lane & 31, lane >> 5
        q_row = w.id * 32 + l31
        np.add.at(l2, q_row, w.v[asm_emu.parse_reg(tab["l"])[1]].view(np.float32).astype(np.float64))
        for i in range(64):
            dt, e = i >> 4, i & 15
            drow = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
            O2[q_row, drow] = w.v[i].view(np.float32)
    for t in range(t_exit, T):
        O2 += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l2 += fsum(t)
    return O2 / l2[:, None], t_exit, codes, n, lines


def swk(row):
    """the m16 kernel's K-tile chunk XOR: depends on the row PART of a fragment only (bits 0-2 and 4), not on its block (j, u)"""
    return ((row & 7) >> 1) | (((row >> 4) & 1) << 2)


def run_m16_statement(pb, lazy_reads, lazy_dma, t0=1, mutate=None):
    """flash_attn_d64_m16_kernel's statement (scripts/gen_attn_m16.py: the d = 64 attention on v_mfma_f32_16x16x32_bf16) for one
    256-query unit: eight waves x 32 queries = two 16-query blocks per wave, K / V^T tiles of 8 KiB through four-slot rings (one
    1 KiB piece of each per wave and tile; the K tile under the kernel's own chunk XOR swk), O in 32 "+a" operands (block (db, qb) at
    4 (2 db + qb)), two running row sums per lane, pre-scaled scores with a zero offset, whole groups of four iterations while
    t + 4 <= tend = T - 3."""
    import gen_attn_m16 as GM
    assert pb.d == 64 and pb.prescaled and not pb.ragged
    lines = GM.emit()
    if mutate is not None:
        lines = mutate(lines)
    T, TILE, NW = pb.T, 8192, 8
    tend = T - 3
    KL, VL = 0, 4 * TILE
    tab = {"o%d" % i: "a%d" % (80 + i) for i in range(32)}
    names_v = ["l0", "l1", "kvo0", "vvo0", "qvo0", "qvo1", "lk0", "lk1", "lv0", "lv1"]
    for i, n in enumerate(names_v):
        tab[n] = "v%d" % i
    for i, n in enumerate(("t", "code", "kstep", "tend", "wk", "wv")):
        tab[n] = "s%d" % i
    for i, n in enumerate(("kb", "vb", "qb")):
        tab[n] = "s[%d:%d]" % (8 + 2 * i, 9 + 2 * i)
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=NW, gmem=pb.gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma)
    qf, kf, vf = pb.q.astype(np.float64), pb.k.astype(np.float64), pb.v.astype(np.float64)
    s_all = qf @ kf.T                                      # log2 units already
    probs = lambda t: bf16_round(np.exp2(s_all[:, t * KVB:(t + 1) * KVB])).astype(np.float64)
    fsum = lambda t: np.exp2(s_all[:, t * KVB:(t + 1) * KVB]).astype(np.float32).astype(np.float64).sum(axis=1)
    O, l = np.zeros((pb.Sq, 64)), np.zeros(pb.Sq)
    for t in range(t0):
        O += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l += fsum(t)

    def stage8(which, tile):
        for wave in range(NW):
            lane = np.arange(64)
            tid = wave * 64 + lane
            row = tid >> 3
            for ln in range(64):
                if which == "k":
                    slot = (tid[ln] & 7) ^ swk(int(row[ln]))
                    src = pb.KOFF + ((tile * KVB + row[ln]) * pb.k_rs + slot * 8) * 2
                    dst = KL + (tile & 3) * TILE + wave * 1024 + ln * 16
                else:
                    slot = (tid[ln] & 7) ^ ((tid[ln] >> 4) & 7)
                    src = pb.VOFF + (row[ln] * pb.vt_rs + slot * 8 + tile * KVB) * 2
                    dst = VL + (tile & 3) * TILE + wave * 1024 + ln * 16
                m.lds[dst:dst + 16] = pb.gmem[src:src + 16]
    for t in range(t0 - 1, t0 + 3):
        stage8("k", t)
    for t in range(t0 - 1, t0 + 2):
        stage8("v", t)

    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        if r[2] == 2:
            w.s[r[1] + 1] = np.uint32(int(val) >> 32)

    def vset(w, name, arr):
        a = np.asarray(arr)
        w.v[asm_emu.parse_reg(tab[name])[1]] = a.view(np.uint32) if a.dtype == np.float32 else a.astype(np.int64).astype(np.uint32)
    for w in m.waves:
        lane = np.arange(64)
        r15, g4, tid = lane & 15, lane >> 4, w.id * 64 + lane
        srow = tid >> 3
        sset(w, "t", t0), sset(w, "tend", tend), sset(w, "kstep", KVB * pb.k_rs * 2)
        sset(w, "wk", KL + w.id * 1024), sset(w, "wv", VL + w.id * 1024)
        sset(w, "kb", pb.KOFF), sset(w, "vb", pb.VOFF), sset(w, "qb", pb.QOFF)
        for qb in range(2):
            q_row = w.id * 32 + 16 * qb + r15
            vset(w, "qvo%d" % qb, (q_row * pb.q_rs + g4 * 8) * 2)
            vset(w, "l%d" % qb, np.where(g4 == 0, l[q_row], 0.0).astype(np.float32))
        rowpart = r15 + 8 * (r15 >> 3)
        for ks in range(2):
            vset(w, "lk%d" % ks, KL + rowpart * 128 + (((4 * ks + g4) ^ swk(rowpart)) * 16))
            vset(w, "lv%d" % ks, VL + r15 * 128 + (((4 * ks + g4) ^ ((r15 >> 1) & 7)) * 16))
        vset(w, "kvo0", (((t0 + 3) * KVB + srow) * pb.k_rs + ((tid & 7) ^ swk(srow)) * 8) * 2)
        vset(w, "vvo0", (srow * pb.vt_rs + ((tid & 7) ^ ((tid >> 4) & 7)) * 8 + (t0 + 2) * KVB) * 2)
        for i in range(32):
            blk, e = i >> 2, i & 3
            db, qb = blk >> 1, blk & 1
            w.a[80 + i] = O[w.id * 32 + 16 * qb + r15, 16 * db + 4 * g4 + e].astype(np.float32).view(np.uint32)
    n = m.run()
    t_exit = int(m.waves[0].s[asm_emu.parse_reg(tab["t"])[1]])
    codes = [int(w.s[asm_emu.parse_reg(tab["code"])[1]]) for w in m.waves]
    O2, l2 = np.zeros((pb.Sq, 64)), np.zeros(pb.Sq)
    for w in m.waves:
        lane = np.arange(64)
        r15, g4 = lane & 15, lane >> 4
        for qb in range(2):
            q_row = w.id * 32 + 16 * qb + r15
            np.add.at(l2, q_row, w.v[asm_emu.parse_reg(tab["l%d" % qb])[1]].view(np.float32).astype(np.float64))
        for i in range(32):
            blk, e = i >> 2, i & 3
            db, qb = blk >> 1, blk & 1
            O2[w.id * 32 + 16 * qb + r15, 16 * db + 4 * g4 + e] = w.a[80 + i].view(np.float32)
    for t in range(t_exit, T):
        O2 += probs(t) @ vf[t * KVB:(t + 1) * KVB]
        l2 += fsum(t)
    return O2 / l2[:, None], t_exit, codes, n, lines
