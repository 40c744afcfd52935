#!/bin/bash
# how often does the first forward of a fresh process differ from the later ones, and where?
mkdir -p gpurun_out; rm -f gpurun_out/c5_rate.log
N=${N:-40}
bad=0
for i in $(seq $N); do
  out=$(env ${ENVX:-XX=1} python scripts/experiments/c5_determinism.py 2 ${FP8:-1} 2>&1 | tail -1)
  case "$out" in *"mismatching runs []"*) ;; *) bad=$((bad+1)); echo "run $i: $out" | cut -c1-900 >> gpurun_out/c5_rate.log;; esac
done
echo "== ${ENVX:-default}: $bad of $N processes had a differing first forward" >> gpurun_out/c5_rate.log
cat gpurun_out/c5_rate.log
