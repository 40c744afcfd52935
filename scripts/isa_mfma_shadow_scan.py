#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -save-temps, *.s) for the hazard behind round 4's attention128_q64 finding:

An MFMA writes its destination registers when its passes are done -- 32+ cycles after issue.  hipcc pads the XDL-write ->
VALU-read / -write hazards of the MFMAs it can see, but an MFMA inside inline-asm text is invisible to it: if such an MFMA's
destination is a compiler-allocated VGPR block (an "=v" / "+v" operand) and the VALUE IS DEAD (never read), hipcc recycles the
registers as temporaries right behind the asm statement, and the MFMA's late write lands on a live temporary.

For every v_mfma whose destination is a VGPR block the scan looks at the instructions up to the next MFMA (at most `window`)
and reports vector / memory instructions that touch a register of that block with fewer than `min_nops` wait states of s_nop
in between.  MFMAs the compiler sees are followed by its own s_nop padding and pass; an unpadded touch is the bug.

    python scripts/isa_mfma_shadow_scan.py file.s [min_nops=4] [window=10] [span=1]   -> exit 1 when something is found
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(path, min_nops=4, window=10, span=1):
    ins, kern = [], None
    for i, line in enumerate(open(path)):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kern = m.group(1)
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ins.append((i + 1, t, kern))
    hits, n_mfma = [], 0
    for k, (ln, t, kn) in enumerate(ins):
        if not t.startswith(("v_mfma", "v_smfmac")):
            continue
        dst = t.split(None, 1)[1].split(",")[0].strip()
        if not dst.startswith("v"):
            continue                      # AccVGPR destinations are named in asm text / clobber lists, never compiler temporaries
        n_mfma += 1
        d, nops, seen = regs(dst), 0, 0
        for ln2, t2, _ in ins[k + 1:k + 1 + window * span]:
            if t2.startswith(("v_mfma", "v_smfmac")):
                seen += 1                 # span = 2: also the instructions behind the NEXT MFMA (its issue waits for the pipe,
                if seen >= span:          # i.e. for this one's last pass -- whose write-back may still be a few cycles out)
                    break
                if regs(t2.split(None, 1)[1].split(",")[0]) == d:
                    break                 # the same block accumulates again: ordered by the pipe
                continue
            if t2.startswith("s_nop"):
                nops += int(t2.split()[1]) + 1
            elif t2.startswith(("v_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
                if regs(t2.split(None, 1)[1] if " " in t2 else "") & d and nops < min_nops:
                    hits.append((kn, ln, t, ln2, t2, nops))
    return n_mfma, hits


if __name__ == "__main__":
    n, hits = scan(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 10,
                   int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    for kn, ln, t, ln2, t2, nops in hits[:40]:
        print("%s\n  %d: %s\n  %d: %s   (%d wait states of s_nop in between)" % (kn, ln, t[:90], ln2, t2[:90], nops))
    print("%s: %d MFMAs with a VGPR destination, %d unpadded touches in their shadow" % (sys.argv[1], n, len(hits)))
    sys.exit(1 if hits else 0)
