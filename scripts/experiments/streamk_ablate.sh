#!/bin/bash
# where the stream-K tail's time goes: 64 = parts not parked, 128 = tiles never completed (no part reads, no epilogue),
# 256 = no acquire fence, 512 = no part reads
mkdir -p gpurun_out
for cfg in "0 0" "1 0" "1 128" "1 192" "1 256" "1 512" "1 768"; do
  set -- $cfg
  echo "== ALG_GEMM_STREAMK=$1 ALG_GEMM_ABLATE=$2" >> gpurun_out/sk_abl.log
  ALG_GEMM_STREAMK=$1 ALG_GEMM_ABLATE=$2 python scripts/kbench.py --only gemm_qk,gemm_ff2 --iters 20 2>&1 | grep "TFLOP/s$" >> gpurun_out/sk_abl.log
done
cat gpurun_out/sk_abl.log
