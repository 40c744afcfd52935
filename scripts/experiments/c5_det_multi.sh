#!/bin/bash
rm -f gpurun_out/c5_rate_all.log
python -m pytest tests/test_gpu_wan_kernels.py -q -k "q64 or d128 or attn" 2>&1 | tail -2
python scripts/kbench.py --only attn128 --iters 7 2>&1 | grep "TFLOP/s$"
ENVX="XX=global_load_lds" N=${N:-90} bash scripts/experiments/c5_det_rate.sh > /dev/null 2>&1
cat gpurun_out/c5_rate.log >> gpurun_out/c5_rate_all.log
grep "^==" gpurun_out/c5_rate_all.log; grep -v "^==" gpurun_out/c5_rate_all.log | cut -c1-330 | head -12
