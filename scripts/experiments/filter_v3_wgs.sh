#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/v3_wgs.log
for w in 0 1 2 3 4 6 8; do
  echo "== WGS=$w" >> gpurun_out/v3_wgs.log
  ALG_LOWPASS_V3_WGS=$w python - >> gpurun_out/v3_wgs.log 2>&1 <<PY
import torch, bench
r = bench.filter_microbench(torch.device("cuda:0"))
print("  ".join(f"{k.replace('down_up','du').replace('gaussian','g').replace('_f32','').replace('videos','v')}:{v['ms']*1e3:.1f}" for k, v in r.items()))
PY
done
grep -v amdgpu.ids gpurun_out/v3_wgs.log
