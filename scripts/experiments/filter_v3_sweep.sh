#!/bin/bash
# few-plane cases and thread counts of the register-blocked filters
mkdir -p gpurun_out; rm -f gpurun_out/v3_sweep.log
for cfg in "513 0" "1 0" "1 256" "1 512" "1 1024" "513 256" "513 512" "513 1024"; do
  set -- $cfg
  echo "== MIN_PLANES=$1 THREADS=$2" >> gpurun_out/v3_sweep.log
  ALG_LOWPASS_V3_MIN_PLANES=$1 ALG_LOWPASS_V3_THREADS=$2 python - >> gpurun_out/v3_sweep.log 2>&1 <<PY
import torch, bench
r = bench.filter_microbench(torch.device("cuda:0"))
print("  ".join(f"{k.replace('down_up','du').replace('gaussian','g').replace('_f32','').replace('videos','v')}:{v['ms']*1e3:.1f}" for k, v in r.items()))
PY
done
grep -v amdgpu.ids gpurun_out/v3_sweep.log
