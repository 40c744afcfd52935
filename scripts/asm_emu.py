#!/usr/bin/env python3
"""A small functional emulator for the gfx950 instruction subset the generated attention statements use (TEST INFRASTRUCTURE: the
CPU suite runs the generated asm text of scripts/gen_attn_q64.py through it and compares the attention it computes with numpy).

What it models: a workgroup of W waves x 64 lanes, ArchVGPRs / AccVGPRs / SGPRs / VCC / SCC / M0 per wave, one LDS, one flat
global memory, s_barrier across the waves, and -- the part that matters for hand-placed waits -- MEMORY ORDERING under the
weakest timing the ISA allows:
    lazy_reads   a ds_read / global_load writes its destination registers only when an s_waitcnt of the issuing wave covers it
                 (lgkmcnt / vmcnt count in issue order); until then the registers keep their OLD content.  An MFMA that consumes
                 a fragment in front of its counted wait computes garbage -> the comparison fails.  The LDS bytes are sampled at
                 completion time (a DMA that overwrites a slot too early corrupts the fragment).
    lazy_dma     a global_load_lds piece lands in LDS only when a vmcnt wait of the issuing wave covers it; other waves see it
                 behind the next barrier.  A fragment read that is not ordered behind (wait, barrier) sees the OLD slot content.
    eager_*      the opposite extreme: everything completes at issue.
A schedule is accepted when it computes the right answer under lazy reads + eager DMA, eager reads + lazy DMA and lazy + lazy.
MFMA result latency (XDL write -> VALU read wait states) is NOT modelled here: tests check it statically on the listing.

Instruction subset: v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16, v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, scales 2^0), ds_read_b128, global_load_dwordx4, global_load_lds_dwordx4, buffer_load_dwordx4 .. offen lds, v_exp_f32, v_fma_f32,
v_add_f32, v_mul_f32, v_mov_b32, v_add_u32, v_min_u32, v_cvt_pk_bf16_f32, v_cmp_ngt_f32, v_accvgpr_{read,write,mov}_b32, s_add_u32, s_sub_u32,
s_mov_b32, s_addc_u32, s_cselect_b32, s_lshl_b32 / lshr / and, v_mul_lo_u32, s_cmp_le_u32 / lt / ge / eq / lg, s_branch, s_cbranch_scc1 / scc0 / vccnz / vccz, s_waitcnt, s_barrier, s_nop.
"""
import re

import numpy as np


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(f):
    """round-to-nearest-even, as v_cvt_pk_bf16_f32 (NaN kept quiet)"""
    u = np.asarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    nan = np.isnan(np.asarray(f, dtype=np.float32))
    return np.where(nan, 0x7FC0, r).astype(np.uint16)


class Wave:
    def __init__(self, wid):
        self.id = wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint32)
        self.vcc = np.zeros(64, dtype=bool)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vm = []      # pending vector-memory ops in issue order: callables
        self.lgkm = []    # pending LDS reads in issue order
        self.n_inst = 0


_RANGE = re.compile(r"^([vas])\[(\d+):(\d+)\]$")
_REG = re.compile(r"^([vas])(\d+)$")


def parse_reg(tok):
    """-> (file, first, count) for v12 / a[4:7] / s[2:3]; None for anything else"""
    m = _RANGE.match(tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
    m = _REG.match(tok)
    if m:
        return m.group(1), int(m.group(2)), 1
    return None


class Machine:
    def __init__(self, text_lines, n_waves=4, lds_bytes=160 * 1024, gmem=None, lazy_reads=True, lazy_dma=True, gmem_va=0):
        self.lines = []
        self.labels = {}
        for ln in text_lines:
            ln = ln.strip()
            if not ln:
                continue
            m = re.match(r"^(\d+):$", ln)
            if m:
                self.labels.setdefault(m.group(1), []).append(len(self.lines))
                continue
            self.lines.append(ln)
        self.waves = [Wave(i) for i in range(n_waves)]
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.gmem = gmem if gmem is not None else np.zeros(0, dtype=np.uint8)
        self.gmem_va = gmem_va            # the virtual address of gmem[0] (tests put a panel across a 4 GiB boundary)
        self.lazy_reads, self.lazy_dma = lazy_reads, lazy_dma
        self.trace = None

    # ---- operand access --------------------------------------------------------------------------------------------
    def rd(self, w, tok):
        """32-bit source operand -> uint32 array [64]"""
        r = parse_reg(tok)
        if r is not None:
            f, i, n = r
            assert n == 1, tok
            if f == "v":
                return w.v[i].copy()
            if f == "a":
                return w.a[i].copy()
            return np.full(64, w.s[i], dtype=np.uint32)
        if tok == "m0":
            return np.full(64, w.m0, dtype=np.uint32)
        if tok.startswith("0x"):
            return np.full(64, int(tok, 16), dtype=np.uint32)
        if re.match(r"^-?\d+$", tok):
            return np.full(64, int(tok) & 0xFFFFFFFF, dtype=np.uint32)
        if re.match(r"^-?\d+\.\d*$", tok):
            return np.full(64, np.float32(float(tok)).view(np.uint32), dtype=np.uint32)
        raise ValueError("operand %r" % tok)

    def rdf(self, w, tok):
        if re.match(r"^-?\d+$", tok):        # an integer inline constant in a float instruction is its float value only for 0
            assert int(tok) == 0, tok
            return np.zeros(64, dtype=np.float32)
        return self.rd(w, tok).view(np.float32)

    def wr(self, w, tok, val):
        f, i, n = parse_reg(tok)
        assert n == 1
        val = np.asarray(val)
        if val.dtype == np.float32:
            val = val.view(np.uint32)
        if f == "v":
            w.v[i] = val
        elif f == "a":
            w.a[i] = val
        else:
            w.s[i] = np.uint32(val if val.ndim == 0 else val[0])

    def tuple_regs(self, w, tok):
        f, i, n = parse_reg(tok)
        return (w.v if f == "v" else w.a), i, n

    def s64(self, w, tok):
        f, i, n = parse_reg(tok)
        assert f == "s" and n == 2, tok
        return int(w.s[i]) | (int(w.s[i + 1]) << 32)

    def gbytes(self, addr):
        """16 bytes per lane from virtual addresses addr [64]; anything outside the image is a memory fault"""
        idx = np.asarray(addr, dtype=np.int64)[:, None] + np.arange(16)[None, :] - self.gmem_va
        if idx.min() < 0 or idx.max() >= self.gmem.size:
            raise RuntimeError("memory fault: global address outside the image")
        return self.gmem[idx]

    # ---- memory completion -------------------------------------------------------------------------------------------
    def drain(self, queue, leave):
        while len(queue) > leave:
            queue.pop(0)()

    # ---- one instruction -------------------------------------------------------------------------------------------
    def step(self, w):
        ln = self.lines[w.pc]
        w.pc += 1
        w.n_inst += 1
        op, _, rest = ln.partition(" ")
        if op == "s_nop":
            return
        if op == "s_waitcnt":
            for m in re.finditer(r"(vmcnt|lgkmcnt)\((\d+)\)", rest):
                self.drain(w.vm if m.group(1) == "vmcnt" else w.lgkm, int(m.group(2)))
            return
        if op == "v_mfma_scale_f32_32x32x64_f8f6f4":
            assert rest.endswith(" op_sel_hi:[0,0,0]"), ln
            self.mfma_f8(w, [a.strip() for a in rest[:-len(" op_sel_hi:[0,0,0]")].split(",")])
            return
        args = [a.strip() for a in rest.split(",")] if rest else []
        mods = {}
        if args and " " in args[-1]:            # trailing modifiers: "v1 offset:32"
            parts = args[-1].split()
            args[-1] = parts[0]
            for p in parts[1:]:
                k, colon, val = p.partition(":")
                mods[k] = int(val, 0) if colon else True          # "offset:32" | flags such as "offen", "lds"
        if op == "s_barrier":
            w.at_barrier = True
            return
        if op == "s_branch" or op.startswith("s_cbranch"):
            take = {"s_branch": True, "s_cbranch_scc1": w.scc == 1, "s_cbranch_scc0": w.scc == 0,
                    "s_cbranch_vccnz": bool(w.vcc.any()), "s_cbranch_vccz": not bool(w.vcc.any())}[op]
            if take:
                m = re.match(r"^(\d+)([bf])$", args[0])
                here = w.pc - 1
                cands = self.labels[m.group(1)]
                w.pc = max(c for c in cands if c <= here) if m.group(2) == "b" else min(c for c in cands if c > here)
            return
        if op == "s_cselect_b32":
            self.wr(w, args[0], np.uint32(int(self.rd(w, args[1] if w.scc else args[2])[0])))
            return
        if op in ("s_add_u32", "s_sub_u32", "s_addc_u32"):
            a, b = int(self.rd(w, args[1])[0]), int(self.rd(w, args[2])[0])
            r = a - b if op == "s_sub_u32" else a + b + (w.scc if op == "s_addc_u32" else 0)
            w.scc = int(r > 0xFFFFFFFF or r < 0)
            if args[0] == "m0":
                w.m0 = r & 0xFFFFFFFF
            else:
                self.wr(w, args[0], np.uint32(r & 0xFFFFFFFF))
            return
        if op == "s_mov_b32":
            val = int(self.rd(w, args[1])[0])
            if args[0] == "m0":
                w.m0 = val
            else:
                self.wr(w, args[0], np.uint32(val))
            return
        if op in ("s_cmp_le_u32", "s_cmp_lt_u32", "s_cmp_ge_u32", "s_cmp_eq_u32", "s_cmp_lg_u32"):
            a, b = int(self.rd(w, args[0])[0]), int(self.rd(w, args[1])[0])
            w.scc = int({"s_cmp_le_u32": a <= b, "s_cmp_lt_u32": a < b, "s_cmp_ge_u32": a >= b, "s_cmp_eq_u32": a == b,
                         "s_cmp_lg_u32": a != b}[op])
            return
        if op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32"):
            a, b = int(self.rd(w, args[1])[0]), int(self.rd(w, args[2])[0])
            r = {"s_lshl_b32": (a << (b & 31)) & 0xFFFFFFFF, "s_lshr_b32": a >> (b & 31), "s_and_b32": a & b}[op]
            w.scc = int(r != 0)
            self.wr(w, args[0], np.uint32(r))
            return
        if op == "v_mul_lo_u32":
            self.wr(w, args[0], (self.rd(w, args[1]).astype(np.uint64) * self.rd(w, args[2]).astype(np.uint64)).astype(np.uint32))
            return
        if op == "v_mov_b32" or op.startswith("v_accvgpr_"):
            self.wr(w, args[0], self.rd(w, args[1]))
            return
        if op == "v_add_u32":
            self.wr(w, args[0], (self.rd(w, args[1]).astype(np.uint64) + self.rd(w, args[2])).astype(np.uint32))
            return
        if op == "v_min_u32":
            self.wr(w, args[0], np.minimum(self.rd(w, args[1]), self.rd(w, args[2])))
            return
        if op in ("v_add_f32", "v_mul_f32"):
            with np.errstate(all="ignore"):
                a, b = self.rdf(w, args[1]), self.rdf(w, args[2])
                self.wr(w, args[0], (a + b if op == "v_add_f32" else a * b).astype(np.float32))
            return
        if op == "v_fma_f32":
            with np.errstate(all="ignore"):
                r = self.rdf(w, args[1]).astype(np.float64) * self.rdf(w, args[2]).astype(np.float64) + self.rdf(w, args[3]).astype(np.float64)
                self.wr(w, args[0], r.astype(np.float32))
            return
        if op == "v_exp_f32":
            with np.errstate(all="ignore"):
                self.wr(w, args[0], np.exp2(self.rdf(w, args[1]).astype(np.float64)).astype(np.float32))
            return
        if op == "v_cvt_pk_bf16_f32":
            lo, hi = f32_to_bf16(self.rdf(w, args[1])), f32_to_bf16(self.rdf(w, args[2]))
            self.wr(w, args[0], lo.astype(np.uint32) | (hi.astype(np.uint32) << 16))
            return
        if op == "v_cmp_ngt_f32":
            assert args[0] == "vcc"
            with np.errstate(all="ignore"):
                w.vcc = ~(self.rdf(w, args[1]) > self.rdf(w, args[2]))
            return
        if op == "v_mfma_f32_32x32x16_bf16":
            self.mfma(w, args)
            return
        if op == "v_mfma_f32_16x16x32_bf16":
            self.mfma16(w, args)
            return
        if op == "ds_read_b128":
            regs, first, n = self.tuple_regs(w, args[0])
            assert n == 4
            addr = (self.rd(w, args[1]).astype(np.int64) + mods.get("offset", 0))

            def complete(regs=regs, first=first, addr=addr):
                idx = addr[:, None] + np.arange(16)[None, :]
                data = self.lds[idx].reshape(64, 4, 4)
                words = data[:, :, 0].astype(np.uint32) | (data[:, :, 1].astype(np.uint32) << 8) | \
                    (data[:, :, 2].astype(np.uint32) << 16) | (data[:, :, 3].astype(np.uint32) << 24)
                for k in range(4):
                    regs[first + k] = words[:, k]
            if self.lazy_reads:
                w.lgkm.append(complete)
            else:
                complete()
            return
        if op == "global_load_dwordx4":
            regs, first, n = self.tuple_regs(w, args[0])
            addr = self.rd(w, args[1]).astype(np.int64) + self.s64(w, args[2]) + mods.get("offset", 0)

            def complete(regs=regs, first=first, addr=addr):
                data = self.gbytes(addr).reshape(64, 4, 4)
                words = data[:, :, 0].astype(np.uint32) | (data[:, :, 1].astype(np.uint32) << 8) | \
                    (data[:, :, 2].astype(np.uint32) << 16) | (data[:, :, 3].astype(np.uint32) << 24)
                for k in range(4):
                    regs[first + k] = words[:, k]
            if self.lazy_reads:
                w.vm.append(complete)
            else:
                complete()
            return
        if op == "global_load_lds_dwordx4":
            addr = self.rd(w, args[0]).astype(np.int64) + self.s64(w, args[1]) + mods.get("offset", 0)
            dst = int(w.m0) + mods.get("offset", 0) + np.arange(64, dtype=np.int64) * 16
            src = self.gbytes(addr).copy()     # global data is read-only here: sample at issue

            def complete(dst=dst, src=src):
                if dst.min() < 0 or dst.max() + 16 > self.lds.size:
                    raise RuntimeError("memory fault: LDS-DMA destination outside the LDS")
                self.lds[dst[:, None] + np.arange(16)[None, :]] = src
            if self.lazy_dma:
                w.vm.append(complete)
            else:
                w.vm.append(lambda: None)       # keeps the vmcnt bookkeeping identical in both modes
                complete()
            return
        if op == "buffer_load_dwordx4" and mods.get("offen") and not mods.get("lds"):
            # 16 bytes per lane into four VGPRs through a raw buffer descriptor; range check on the per-lane offset + immediate
            regs, first, n = self.tuple_regs(w, args[0])
            assert n == 4
            f, i, nn = parse_reg(args[2])
            assert f == "s" and nn == 4 and i % 4 == 0, args[2]
            base = (int(w.s[i]) | ((int(w.s[i + 1]) & 0xFFFF) << 32)) + int(self.rd(w, args[3])[0])
            num = int(w.s[i + 2])
            off = self.rd(w, args[1]).astype(np.int64) + mods.get("offset", 0)
            inb = off + 16 <= num
            data = self.gbytes(np.where(inb, base + off, self.gmem_va)).copy()
            data[~inb] = 0

            def complete(regs=regs, first=first, data=data.reshape(64, 4, 4)):
                words = data[:, :, 0].astype(np.uint32) | (data[:, :, 1].astype(np.uint32) << 8) | \
                    (data[:, :, 2].astype(np.uint32) << 16) | (data[:, :, 3].astype(np.uint32) << 24)
                for k in range(4):
                    regs[first + k] = words[:, k]
            if self.lazy_dma:
                w.vm.append(complete)
            else:
                w.vm.append(lambda: None)
                complete()
            return
        if op == "buffer_load_dwordx4" and mods.get("lds") and mods.get("offen"):
            # LDS-DMA through a raw buffer descriptor s[i:i+3] = (base lo, base hi & 0xffff | stride 0, num_records, flags):
            # address = base + scalar offset + per-lane offset.  Range check as the hardware does it for raw buffers: ONLY the
            # per-lane offset (+ the instruction offset) is compared with num_records -- the scalar offset is not -- and a lane
            # out of range fetches nothing and stores zeros
            f, i, n = parse_reg(args[1])
            assert f == "s" and n == 4 and i % 4 == 0, args[1]
            base = int(w.s[i]) | ((int(w.s[i + 1]) & 0xFFFF) << 32)
            assert (int(w.s[i + 1]) >> 16) == 0, "stride / swizzle bits must be clear"
            num = int(w.s[i + 2])
            off = self.rd(w, args[0]).astype(np.int64) + mods.get("offset", 0)
            base += int(self.rd(w, args[2])[0])
            inb = off + 16 <= num
            src = self.gbytes(np.where(inb, base + off, self.gmem_va)).copy()
            src[~inb] = 0
            dst = int(w.m0) + mods.get("offset", 0) + np.arange(64, dtype=np.int64) * 16

            def complete(dst=dst, src=src):
                if dst.min() < 0 or dst.max() + 16 > self.lds.size:
                    raise RuntimeError("memory fault: LDS-DMA destination outside the LDS")
                self.lds[dst[:, None] + np.arange(16)[None, :]] = src
            if self.lazy_dma:
                w.vm.append(complete)
            else:
                w.vm.append(lambda: None)
                complete()
            return
        raise NotImplementedError(ln)

    E4M3 = None

    @classmethod
    def e4m3_table(cls):
        """OCP e4m3 (fn: no infinities, 0x7f / 0xff = NaN) byte -> value"""
        if cls.E4M3 is None:
            t = np.zeros(256, dtype=np.float64)
            for b in range(256):
                e, m = (b >> 3) & 15, b & 7
                v = (m / 8.0) * 2.0 ** -6 if e == 0 else (np.nan if (e == 15 and m == 7) else (1.0 + m / 8.0) * 2.0 ** (e - 7))
                t[b] = -v if b & 0x80 else v
            cls.E4M3 = t
        return cls.E4M3

    def mfma_f8(self, w, args):
        """v_mfma_scale_f32_32x32x64_f8f6f4 with both operands e4m3 and every block scale 2^0 (scale registers 0x7f7f7f7f): D = A (32 x 64)
        * B (64 x 32) + C.  Lane l holds row / column l % 32 and the 32 bytes of k-group l / 32 in eight registers; which 32 of the 64 k
        a group is does not matter to a program that stages A and B the same way (the sum runs over all of them once)."""
        dregs, d0, dn = self.tuple_regs(w, args[0])
        assert dn == 16
        for sc in (args[4], args[5].split()[0]):
            assert (self.rd(w, sc) == 0x7F7F7F7F).all(), "block scales must be 2^0"
        tab = self.e4m3_table()

        def frag(tok):
            regs, f0, n = self.tuple_regs(w, tok)
            assert n == 8
            words = regs[f0:f0 + 8]                                   # [8, 64]
            by = np.stack([(words >> (8 * i)) & 0xFF for i in range(4)], axis=1).reshape(32, 64)   # byte index 4 r + i, lane
            m = np.zeros((32, 64), dtype=np.float64)
            for h2 in range(2):
                m[:, 32 * h2:32 * h2 + 32] = tab[by[:, 32 * h2:32 * h2 + 32]].T
            return m
        A, Bt = frag(args[1]), frag(args[2])
        cregs, c0, cn = self.tuple_regs(w, args[3])
        D = A @ Bt.T
        for e in range(16):
            for h2 in range(2):
                rows = (e & 3) + 8 * (e >> 2) + 4 * h2
                cur = cregs[c0 + e][32 * h2:32 * h2 + 32].view(np.float32).astype(np.float64)
                out = (cur + D[rows, :]).astype(np.float32)
                dregs[d0 + e][32 * h2:32 * h2 + 32] = out.view(np.uint32)

    def mfma(self, w, args):
        """D = A (32 x 16) * B (16 x 32) + C; lane l: A row l % 32, k = 8 (l / 32) + i; B column l % 32, same k; D column l % 32,
        register e <-> row (e & 3) + 8 (e >> 2) + 4 (l / 32)."""
        dregs, d0, dn = self.tuple_regs(w, args[0])
        assert dn == 16

        def frag(tok):
            regs, f0, n = self.tuple_regs(w, tok)
            assert n == 4
            words = regs[f0:f0 + 4]                                   # [4, 64]
            lo, hi = bf16_to_f32((words & 0xFFFF).astype(np.uint16)), bf16_to_f32((words >> 16).astype(np.uint16))
            m = np.zeros((32, 16), dtype=np.float64)
            for h2 in range(2):
                for r in range(4):
                    m[:, 8 * h2 + 2 * r] = lo[r, 32 * h2:32 * h2 + 32]
                    m[:, 8 * h2 + 2 * r + 1] = hi[r, 32 * h2:32 * h2 + 32]
            return m
        A, Bt = frag(args[1]), frag(args[2])                           # A[m][k], Bt[n][k]
        with np.errstate(all="ignore"):
            D = A @ Bt.T                                                 # [m][n]
        if args[3] != "0":
            cregs, c0, cn = self.tuple_regs(w, args[3])
            assert cn == 16
            C = cregs[c0:c0 + 16].view(np.float32)
        else:
            C = np.zeros((16, 64), dtype=np.float32)
        out = np.empty((16, 64), dtype=np.float32)
        for e in range(16):
            for h2 in range(2):
                row = (e & 3) + 8 * (e >> 2) + 4 * h2
                with np.errstate(all="ignore"):
                    out[e, 32 * h2:32 * h2 + 32] = (D[row, :] + C[e, 32 * h2:32 * h2 + 32].astype(np.float64)).astype(np.float32)
        dregs[d0:d0 + 16] = out.view(np.uint32)

    def mfma16(self, w, args):
        """v_mfma_f32_16x16x32_bf16: D = A (16 x 32) * B (32 x 16) + C; lane l: A row l % 16, k = 8 (l / 16) + i; B column l % 16,
        same k; D column l % 16, register e <-> row 4 (l / 16) + e."""
        dregs, d0, dn = self.tuple_regs(w, args[0])
        assert dn == 4

        def frag(tok):
            regs, f0, n = self.tuple_regs(w, tok)
            assert n == 4
            words = regs[f0:f0 + 4]                                   # [4, 64]
            lo, hi = bf16_to_f32((words & 0xFFFF).astype(np.uint16)), bf16_to_f32((words >> 16).astype(np.uint16))
            m = np.zeros((16, 32), dtype=np.float64)
            for g in range(4):
                for r in range(4):
                    m[:, 8 * g + 2 * r] = lo[r, 16 * g:16 * g + 16]
                    m[:, 8 * g + 2 * r + 1] = hi[r, 16 * g:16 * g + 16]
            return m
        A, Bt = frag(args[1]), frag(args[2])                           # A[m][k], Bt[n][k]
        with np.errstate(all="ignore"):
            D = A @ Bt.T                                                 # [m][n]
        if args[3] != "0":
            cregs, c0, cn = self.tuple_regs(w, args[3])
            assert cn == 4
            C = cregs[c0:c0 + 4].view(np.float32)
        else:
            C = np.zeros((4, 64), dtype=np.float32)
        out = np.empty((4, 64), dtype=np.float32)
        for e in range(4):
            for g in range(4):
                with np.errstate(all="ignore"):
                    out[e, 16 * g:16 * g + 16] = (D[4 * g + e, :] + C[e, 16 * g:16 * g + 16].astype(np.float64)).astype(np.float32)
        dregs[d0:d0 + 4] = out.view(np.uint32)

    # ---- run ---------------------------------------------------------------------------------------------------------
    def run(self, max_inst=10_000_000):
        total = 0
        while True:
            progressed = False
            for w in self.waves:
                while not w.done and not w.at_barrier:
                    if w.pc >= len(self.lines):
                        w.done = True
                        self.drain(w.vm, 0)
                        self.drain(w.lgkm, 0)
                        break
                    self.step(w)
                    progressed = True
                    total += 1
                    if total > max_inst:
                        raise RuntimeError("instruction budget exceeded (runaway loop?)")
            live = [w for w in self.waves if not w.done]
            if not live:
                return total
            if all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                progressed = True
            if not progressed:
                raise RuntimeError("deadlock: some waves wait at a barrier that others never reach")


def bind(lines, table):
    """replace %[name] by the register string table[name]"""
    out = []
    for ln in lines:
        out.append(re.sub(r"%\[(\w+)\]", lambda m: table[m.group(1)], ln))
    return out
