#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/c5_det.log
run() { echo "== $*" >> gpurun_out/c5_det.log; env "$@" python scripts/experiments/c5_determinism.py 8 1 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/c5_det.log; }
run ALG_X=0
run ALG_ATTN128_Q64=0
run ALG_GEMM_PIPE=0
run ALG_X=1
echo "== bf16 weights" >> gpurun_out/c5_det.log; python scripts/experiments/c5_determinism.py 8 0 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/c5_det.log
cat gpurun_out/c5_det.log
