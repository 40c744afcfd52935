"""Harness that runs the generated GEMM schedule-9 main-loop statement (scripts/gen_gemm_p9.py -> gemm_p9_loop.inc, the plain
form) in the instruction-level emulator (scripts/asm_emu.py) the way gemm_kernel.h's frame drives it: one 256 x 256 tile, four
waves (2 x 2 of 128 x 128), A and B row-major panels in a NaN-filled memory image, the ten-slot LDS ring, 256 accumulators per
lane.  Test infrastructure (only tests/ import it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import asm_emu  # noqa: E402
import gen_gemm_p9 as GP  # noqa: E402
import gen_gemm_p10 as GP10  # noqa: E402


def bf16_round(x):
    return asm_emu.bf16_to_f32(asm_emu.f32_to_bf16(np.asarray(x, dtype=np.float32)))


class ProblemFp8:
    """the same tile with OCP e4m3 operands: a k-tile is 128 bytes = 128 values per row (K = 128 nk)"""

    def __init__(self, nk, seed=0, rows_a=256, rows_b=256):
        rng = np.random.default_rng(seed)
        self.K, self.nk, self.rows_a, self.rows_b = 128 * nk, nk, rows_a, rows_b
        self.lda, self.ldb, self.ldr = self.K, self.K + 128, 264           # (pitches in ELEMENTS = bytes here)
        tab = asm_emu.Machine.e4m3_table()
        codes = lambda n: np.where((c := rng.integers(0, 256, size=(n, self.K), dtype=np.uint8)) & 0x7F == 0x7F, 0x38, c).astype(np.uint8)
        self.ca, self.cb = codes(rows_a), codes(rows_b)
        self.a, self.b = tab[self.ca], tab[self.cb]
        self.r = bf16_round(rng.standard_normal((rows_a, 256)))
        self.AOFF = 4096
        self.BOFF = self.AOFF + rows_a * self.lda + 640
        self.ROFF = self.BOFF + rows_b * self.ldb + 384
        g = np.full(self.ROFF + 2 * rows_a * self.ldr + 256, 0x7F, dtype=np.uint8)                # e4m3 NaN everywhere else
        for r in range(rows_a):
            g[self.AOFF + r * self.lda: self.AOFF + r * self.lda + self.K] = self.ca[r]
            g[self.ROFF + 2 * r * self.ldr: self.ROFF + 2 * r * self.ldr + 512] = asm_emu.f32_to_bf16(self.r[r]).view(np.uint8)
        for r in range(rows_b):
            g[self.BOFF + r * self.ldb: self.BOFF + r * self.ldb + self.K] = self.cb[r]
        self.gmem = g
        self.esz = 1

    def reference(self):
        a = np.zeros((256, self.K)); a[:self.rows_a] = self.a; a[self.rows_a:] = self.a[-1]
        b = np.zeros((256, self.K)); b[:self.rows_b] = self.b; b[self.rows_b:] = self.b[-1]
        return a @ b.T


class Problem:
    esz = 2

    def __init__(self, nk, seed=0, rows_a=256, rows_b=256, lda=None, ldb=None):
        """one tile: C[256, 256] = A[rows_a, K] B[rows_b, K]^T, K = 64 nk; rows past rows_a / rows_b do not exist (the frame clamps
        the DMA's row index to the last valid one: rmaxa / rmaxb)"""
        rng = np.random.default_rng(seed)
        self.K, self.nk = 64 * nk, nk
        self.rows_a, self.rows_b = rows_a, rows_b
        self.lda, self.ldb = lda or self.K, ldb or self.K + 64
        self.a = bf16_round(rng.standard_normal((rows_a, self.K)))
        self.b = bf16_round(rng.standard_normal((rows_b, self.K)))
        self.ldr = 256 + 8
        self.r = bf16_round(rng.standard_normal((rows_a, 256)))      # residual [M = rows_a, N = 256]
        self.AOFF = 4096
        self.BOFF = self.AOFF + 2 * rows_a * self.lda + 640
        self.ROFF = self.BOFF + 2 * rows_b * self.ldb + 384
        size = self.ROFF + 2 * rows_a * self.ldr + 256
        g16 = np.full(size // 2, 0x7FC0, dtype=np.uint16)           # NaN everywhere outside the panels' elements
        for r in range(rows_a):
            g16[self.ROFF // 2 + r * self.ldr: self.ROFF // 2 + r * self.ldr + 256] = asm_emu.f32_to_bf16(self.r[r])
        for r in range(rows_a):
            g16[self.AOFF // 2 + r * self.lda: self.AOFF // 2 + r * self.lda + self.K] = asm_emu.f32_to_bf16(self.a[r])
        for r in range(rows_b):
            g16[self.BOFF // 2 + r * self.ldb: self.BOFF // 2 + r * self.ldb + self.K] = asm_emu.f32_to_bf16(self.b[r])
        self.gmem = g16.view(np.uint8)

    def reference(self):
        a = np.zeros((256, self.K)); a[:self.rows_a] = self.a; a[self.rows_a:] = self.a[-1]     # clamped rows repeat the last one
        b = np.zeros((256, self.K)); b[:self.rows_b] = self.b; b[self.rows_b:] = self.b[-1]
        return a.astype(np.float64) @ b.astype(np.float64).T


def run_plain(pb, lazy_reads, lazy_dma, mutate=None, va=(1 << 32) - 70000, res=False, sched=9):
    """-> C [256, 256] float64 as the statement leaves it in the accumulators, instruction count (res: the residual form -- also
    returns the residual tile as the statement hands it to the epilogue, [256, 256] float64).  sched = 10: the 16x16x32 statement
    (scripts/gen_gemm_p10.py): other fragment addresses, other (register, lane) -> (row, column) map, same frame otherwise."""
    if pb.esz == 1:
        import gen_gemm_p9_fp8 as GF
        lines = GF.emit(res)
    elif sched == 10:
        lines = GP10.emit(res=res)
    else:
        lines = GP.emit(res=res)
    if mutate is not None:
        lines = mutate(lines)
    tab = {}
    if res:
        for i in range(32):
            tab["r%d" % i] = "v[%d:%d]" % (16 + 4 * i, 19 + 4 * i)
        tab["rvoff"], tab["ldr16"], tab["rs"] = "v6", "s18", "s[24:27]"
    for i, n in enumerate(("vl0", "vl1", "vl2", "vl3", "vrow", "vslot")):
        tab[n] = "v%d" % i
    for i in range(10):
        tab["t%d" % i] = "s%d" % i
    for i, n in enumerate(("lda2", "ldb2", "rmaxa", "rmaxb", "nloop", "wm", "wn2", "wave1k")):
        tab[n] = "s%d" % (10 + i)
    tab["pa"], tab["pb"] = "s[20:21]", "s[22:23]"
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=4, gmem=pb.gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma, gmem_va=va)
    SMEM = 0

    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        if r[2] == 2:
            w.s[r[1] + 1] = np.uint32(int(val) >> 32)
    for w in m.waves:
        ln = np.arange(64)
        l31, h2 = ln & 31, ln >> 5
        sw = (l31 >> 1) & 7
        wm, wn = w.id // 2, w.id % 2
        w.v[4] = (w.id * 8 + (ln >> 3)).astype(np.uint32)                                     # vrow
        w.v[5] = (((ln & 7) ^ ((w.id * 4 + (ln >> 4)) & 7)) * 16).astype(np.uint32)           # vslot
        for ks in range(4):
            w.v[ks] = (l31 * 128 + (((2 * ks + h2) ^ sw) * 16)).astype(np.uint32)             # vl0..3
        if sched == 10:
            l15, g4 = ln & 15, ln >> 4
            for ks in range(2):
                w.v[ks] = (l15 * 128 + (((4 * ks + g4) ^ ((l15 >> 1) & 7)) * 16)).astype(np.uint32)
            w.v[2] = w.v[3] = np.full(64, 0xDEAD0000, dtype=np.uint32)                        # unused by schedule 10
        sset(w, "lda2", pb.lda * pb.esz), sset(w, "ldb2", pb.ldb * pb.esz)
        sset(w, "rmaxa", pb.rows_a - 1), sset(w, "rmaxb", pb.rows_b - 1)
        sset(w, "nloop", pb.nk - 2), sset(w, "wm", wm), sset(w, "wn2", 2 + wn), sset(w, "wave1k", SMEM + w.id * 1024)
        sset(w, "pa", va + pb.AOFF), sset(w, "pb", va + pb.BOFF)
        if res:
            # the frame's descriptor: num_records ends R behind its last row, so rows past M read as zeros; the lane's whole byte
            # offset (tile origin = 0 here) rides in the vector offset
            w.v[6] = (((wm * 128 + (ln >> 2)) * pb.ldr + wn * 128 + (ln & 3) * 8) * 2).astype(np.uint32)
            sset(w, "ldr16", pb.ldr * 32)
            base = va + pb.ROFF
            for i, val in enumerate((base & 0xFFFFFFFF, (base >> 32) & 0xFFFF, ((pb.rows_a - 1) * pb.ldr + 256) * 2, 0x00020000)):
                w.s[24 + i] = np.uint32(val)
    n = m.run()
    C = np.zeros((256, 256))
    for w in m.waves:
        ln = np.arange(64)
        l31, h2 = ln & 31, ln >> 5
        wm, wn = w.id // 2, w.id % 2
        for mt in range(4):
            for nt in range(4):
                for e in range(16):
                    # C^T layout: D = MFMA(B fragment, A fragment): lane column = the A row (m), register row = the B row (n)
                    if sched == 10:   # 16 x 16 blocks: quad q = 2 bm + bn of the 32 x 32 region, lane = (m & 15, n quad)
                        q, i = e >> 2, e & 3
                        mrow = wm * 128 + mt * 32 + 16 * (q >> 1) + (ln & 15)
                        ncol = wn * 128 + nt * 32 + 16 * (q & 1) + 4 * (ln >> 4) + i
                    else:
                        mrow = wm * 128 + mt * 32 + l31
                        ncol = wn * 128 + nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2
                    C[mrow, ncol] = w.a[16 * (4 * mt + nt) + e].view(np.float32)
    if not res:
        return C, n, lines
    R = np.zeros((256, 256))
    for w in m.waves:
        ln = np.arange(64)
        wm, wn = w.id // 2, w.id % 2
        for c in range(8):
            for nt in range(4):
                it = (((c >> 1) * 4 + nt) << 1) | (c & 1)
                for k in range(4):
                    word = w.v[16 + 4 * it + k]
                    row, col = wm * 128 + 16 * c + (ln >> 2), wn * 128 + nt * 32 + (ln & 3) * 8 + 2 * k
                    R[row, col] = asm_emu.bf16_to_f32((word & 0xFFFF).astype(np.uint16))
                    R[row, col + 1] = asm_emu.bf16_to_f32((word >> 16).astype(np.uint16))
    return C, R, n, lines


# ---------------------------------------------------------------------------------------------------------------------
# schedule 11 (scripts/gen_gemm_p11.py): 1 x 4 wave layout, A through the LDS ring, B pre-packed in fragment order and loaded
# straight into registers
# ---------------------------------------------------------------------------------------------------------------------
def pack_b_p11(b, nk):
    """what alg_pack_b_p11 writes for ONE 256-column tile: [k-tile][wave 4][k-step 2][n-block 4][lane 64][8 bf16]; rows past the
    operand's last row are zeros"""
    rows = b.shape[0]
    out = np.zeros((nk, 4, 2, 4, 64, 8), dtype=np.uint16)
    bb = asm_emu.f32_to_bf16(b)
    for w in range(4):
        for bj in range(4):
            for lane in range(64):
                n = 64 * w + 16 * bj + (lane & 15)
                if n >= rows:
                    continue
                g = lane >> 4
                for kt in range(nk):
                    for ks in range(2):
                        k0 = 64 * kt + 32 * ks + 8 * g
                        out[kt, w, ks, bj, lane] = bb[n, k0:k0 + 8]
    return out


def run_p11(pb, lazy_reads, lazy_dma, mutate=None, va=(1 << 32) - 70000, res=False):
    """-> C [256, 256] float64 as schedule 11's statement leaves it in the accumulators (+ the residual tile in the residual form)"""
    import gen_gemm_p11 as GP11
    lines = GP11.emit(res=res)
    if mutate is not None:
        lines = mutate(lines)
    # the packed B panel lives behind the problem's own image (A, row-major B -- unused here --, R)
    packed = pack_b_p11(pb.b, pb.nk).reshape(-1).view(np.uint8)
    POFF = (pb.gmem.size + 255) // 256 * 256 + 128
    gmem = np.full(POFF + packed.size + 512, 0xC0, dtype=np.uint8)
    gmem[:pb.gmem.size] = pb.gmem
    gmem[1:POFF:2] = np.where(np.arange(1, POFF, 2) >= pb.gmem.size, 0x7F, gmem[1:POFF:2])      # (bf16 NaN in the gap)
    gmem[POFF:POFF + packed.size] = packed
    gmem[POFF + packed.size:] = 0xFF                                                              # NaN behind the panel
    tab = {}
    if res:
        for i in range(32):
            tab["r%d" % i] = "v[%d:%d]" % (16 + 4 * i, 19 + 4 * i)
        tab["rvoff"], tab["ldr16"], tab["rs"] = "v6", "s18", "s[24:27]"
    for i, n in enumerate(("vl0", "vl1", "vrow", "vslot", "vb0", "vb1")):
        tab[n] = "v%d" % i
    for i in range(10):
        tab["t%d" % i] = "s%d" % i
    for i, n in enumerate(("lda2", "rmaxa", "nloop", "wave1k")):
        tab[n] = "s%d" % (10 + i)
    tab["pa"], tab["db"] = "s[20:21]", "s[28:31]"
    m = asm_emu.Machine(asm_emu.bind(lines, tab), n_waves=4, gmem=gmem, lazy_reads=lazy_reads, lazy_dma=lazy_dma, gmem_va=va)

    def sset(w, name, val):
        r = asm_emu.parse_reg(tab[name])
        w.s[r[1]] = np.uint32(int(val) & 0xFFFFFFFF)
        if r[2] == 2:
            w.s[r[1] + 1] = np.uint32(int(val) >> 32)
    for w in m.waves:
        ln = np.arange(64)
        l15, g4 = ln & 15, ln >> 4
        w.v[2] = (w.id * 8 + (ln >> 3)).astype(np.uint32)                                     # vrow
        w.v[3] = (((ln & 7) ^ ((w.id * 4 + (ln >> 4)) & 7)) * 16).astype(np.uint32)           # vslot
        for ks in range(2):
            w.v[ks] = (l15 * 128 + (((4 * ks + g4) ^ ((l15 >> 1) & 7)) * 16)).astype(np.uint32)
        w.v[4] = (ln * 16).astype(np.uint32)
        w.v[5] = (ln * 16 + 4096).astype(np.uint32)
        sset(w, "lda2", pb.lda * 2), sset(w, "rmaxa", pb.rows_a - 1), sset(w, "nloop", pb.nk - 2), sset(w, "wave1k", w.id * 1024)
        sset(w, "pa", va + pb.AOFF)
        base = va + POFF + w.id * 8192
        for i, val in enumerate((base & 0xFFFFFFFF, (base >> 32) & 0xFFFF, pb.nk * 32768 - w.id * 8192 - 24576 + 8192, 0x00020000)):
            w.s[28 + i] = np.uint32(val)
        if res:
            w.v[6] = (((ln >> 2) * pb.ldr + w.id * 64 + (ln & 3) * 8) * 2).astype(np.uint32)
            sset(w, "ldr16", pb.ldr * 32)
            rb = va + pb.ROFF
            for i, val in enumerate((rb & 0xFFFFFFFF, (rb >> 32) & 0xFFFF, ((pb.rows_a - 1) * pb.ldr + 256) * 2, 0x00020000)):
                w.s[24 + i] = np.uint32(val)
    n = m.run()
    C = np.zeros((256, 256))
    for w in m.waves:
        ln = np.arange(64)
        for mt in range(8):
            for nt in range(2):
                for e in range(16):
                    q, i = e >> 2, e & 3
                    mrow = mt * 32 + 16 * (q >> 1) + (ln & 15)
                    ncol = w.id * 64 + nt * 32 + 16 * (q & 1) + 4 * (ln >> 4) + i
                    C[mrow, ncol] = w.a[16 * (2 * mt + nt) + e].view(np.float32)
    if not res:
        return C, n, lines
    R = np.zeros((256, 256))
    for w in m.waves:
        ln = np.arange(64)
        for mt in range(8):
            for nt in range(2):
                for half in range(2):
                    it = ((mt * 2 + nt) << 1) | half
                    for k in range(4):
                        word = w.v[16 + 4 * it + k]
                        row, col = mt * 32 + 16 * half + (ln >> 2), w.id * 64 + nt * 32 + (ln & 3) * 8 + 2 * k
                        R[row, col] = asm_emu.bf16_to_f32((word & 0xFFFF).astype(np.uint16))
                        R[row, col + 1] = asm_emu.bf16_to_f32((word >> 16).astype(np.uint16))
    return C, R, n, lines
