#!/bin/bash
# A/B of the single-launch stream-K tail (ALG_GEMM_STREAMK=0 vs default) at the C2 GEMM shapes + the parity tests
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dit_kernels.py -q -x -k "stream_k or gemm" 2>&1 | tail -5 > gpurun_out/sk_tests.log
for sk in 0 1 0 1; do
  echo "== ALG_GEMM_STREAMK=$sk" >> gpurun_out/sk_ab.log
  ALG_GEMM_STREAMK=$sk python scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 20 >> gpurun_out/sk_ab.log 2>&1
done
for mr in 8 16 24; do
  echo "== MIN_RUN=$mr" >> gpurun_out/sk_ab.log
  ALG_GEMM_STREAMK_MIN_RUN=$mr python scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 20 >> gpurun_out/sk_ab.log 2>&1
done
cat gpurun_out/sk_tests.log; cat gpurun_out/sk_ab.log
