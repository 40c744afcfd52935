#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for the dominant kernel, the mean duration of the launches
that fall INSIDE bench.py's timed region (the last `launches` of them: warm-up launches come first) next to the figure the bench line's
own HIP events gave (roofline.mean_launch_ms), and the means per grid size (the 2-sample and 3-sample launches of the ALG schedule).
usage: rocprof_timed_region.py <kernel_trace.csv> <bench.json> [kernel-name-substring]"""
import csv
import json
import sys
from collections import defaultdict


def main():
    trace, bench = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else None
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    roof = line["roofline"]
    want = want or roof["kernel"]
    rows = []
    with open(trace, newline="") as f:
        for r in csv.DictReader(f):
            if want in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                             int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)))
    rows.sort()
    n = int(roof["launches"])
    timed = rows[-n:]
    mean = lambda xs: sum(xs) / max(len(xs), 1)
    by_grid = defaultdict(list)
    for _, d, g in timed:
        by_grid[g].append(d)
    out = {"kernel": want, "launches_in_trace": len(rows), "launches_in_timed_region": len(timed),
           "rocprof_mean_ms_timed_region": mean([d for _, d, _ in timed]) / 1e6,
           "rocprof_mean_ms_all_launches": mean([d for _, d, _ in rows]) / 1e6,
           "bench_events_mean_launch_ms": roof["mean_launch_ms"],
           "per_grid_timed_region": {str(g): {"n": len(v), "mean_ms": mean(v) / 1e6} for g, v in sorted(by_grid.items())}}
    out["events_over_rocprof"] = out["bench_events_mean_launch_ms"] / out["rocprof_mean_ms_timed_region"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
