#!/bin/bash
rm -f gpurun_out/c5_rate_all.log
for e in "ALG_ATTN128_SYNC_BEFORE=1 ALG_ATTN128_Q64=0" "ALG_ATTN128_SYNC_BEFORE=1 ALG_ATTN128_Q64=1"; do
  ENVX="$e" N=${N:-45} bash scripts/experiments/c5_det_rate.sh > /dev/null 2>&1
  cat gpurun_out/c5_rate.log >> gpurun_out/c5_rate_all.log
done
grep "^==" gpurun_out/c5_rate_all.log; grep -v "^==" gpurun_out/c5_rate_all.log | cut -c1-330 | head -8
