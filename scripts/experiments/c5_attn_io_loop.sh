#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/c5_io.log
for i in $(seq ${N:-45}); do
  out=$(ALG_ATTN128_Q64=1 python scripts/experiments/c5_attn_io_diff.py 2>&1 | tail -1)
  case "$out" in *"mismatching forwards: [] | none"*) ;; *) echo "run $i: $out" | cut -c1-1800 >> gpurun_out/c5_io.log;; esac
done
echo "done ${N:-45}" >> gpurun_out/c5_io.log; cat gpurun_out/c5_io.log
