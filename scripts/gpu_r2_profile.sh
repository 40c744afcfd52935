#!/bin/bash
# Round-2 evidence run: attention static-priority A/B, rocprofv3 kernel stats of the DEFAULT bench command, the bench itself.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
for rep in 1 2 3; do
  for p in 0 1; do
    echo "== rep $rep ALG_ATTN_PRIO=$p" >> $O/attn_prio.log
    ALG_ATTN_PRIO=$p timeout 300 python scripts/kbench.py --only attn --iters 9 2>&1 | grep -v amdgpu.ids >> $O/attn_prio.log
  done
done
tail -30 $O/attn_prio.log
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd -o r2 -- python $R/bench.py --no-cpu-baseline > $O/profd_bench.json 2> $O/profd.err
echo "profdefault exit $?"
cd $R
find $O/profd -name "*kernel_trace*" -delete; ls -la $O/profd/*; tail -c 400 $O/profd_bench.json
timeout 1500 python bench.py --cross-check --c1-budget 1500 > $O/bench_full.json 2> $O/bench_full.err; echo "bench exit $?"; tail -c 2500 $O/bench_full.json
