#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
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
                acc[k[:60]][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print("==", os.path.basename(d))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("  %-60s %-28s n=%-4d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
