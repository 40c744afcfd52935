#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch.
PMC_BY_GRID=1: per (kernel name, grid size) -- a bench run launches the same kernel for 2-sample and 3-sample steps."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
BY_GRID = os.environ.get("PMC_BY_GRID") == "1"
allc = defaultdict(dict)          # kernel -> counter -> mean, over all passes (for the derived ratios at the end)
for d in sorted(glob.glob(os.path.join(root, "*_p*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                if "alg" not in k:
                    continue
                key = k[:60] + (" [grid=%s]" % row.get("Grid_Size", "?") if BY_GRID else "")
                acc[key][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print("==", os.path.basename(d))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("  %-*s %-28s n=%-4d mean=%.4g" % (80 if BY_GRID else 60, k, c, len(v), sum(v) / len(v)))
            allc[k][c] = sum(v) / len(v)
print("== derived (per kernel; SQ_INSTS_VALU counts every VALU instruction once, MFMAs included -- calibrated below; 1024 SIMDs;")
print("   MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over the launch's shader cycles, taken as GRBM_GUI_ACTIVE / 8 (one instance")
print("   per XCD in this rocprofv3; cross-check: SQ_BUSY_CYCLES / 32 shader engines gives the same cycles to 5 %))")
for k, c in allc.items():
    if c.get("SQ_INSTS_MFMA", 0) > 0:
        m = c["SQ_INSTS_MFMA"]
        line = "  %-60s VALU/MFMA %.2f  SALU/MFMA %.2f  LDS/MFMA %.2f" % (k, c.get("SQ_INSTS_VALU", 0) / m, c.get("SQ_INSTS_SALU", 0) / m,
                                                                          c.get("SQ_INSTS_LDS", 0) / m)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            line += "  MFMA-busy %.1f %% of SIMD cycles" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0))
        print(line)
