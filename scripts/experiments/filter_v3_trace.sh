#!/bin/bash
# kernel-only durations of the filter microbench (rocprofv3 kernel trace): separates the kernels from the host wrapper
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v3trace; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/scripts/filter_bench.py > $O/bench.json 2> $O/err.log
cd $R
f=$(ls $O/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "down_up" in n or "gaussian" in n:
        print(f'{n[:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
