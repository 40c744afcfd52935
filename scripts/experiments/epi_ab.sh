#!/bin/bash
# A/B of the GEMM epilogue with hoisted operand loads: old library (alg_amd/libalg_hip_old.so) vs the current build
mkdir -p gpurun_out; rm -f gpurun_out/epi_ab.log
python -m pytest tests/test_gpu_dit_kernels.py tests/test_gpu_fp8.py -q -x -k "gemm or fp8 or conv" 2>&1 | tail -4 >> gpurun_out/epi_ab.log
for rep in 1 2; do
  for lib in old new; do
    echo "== $lib (rep $rep)" >> gpurun_out/epi_ab.log
    if [ $lib = old ]; then export ALG_HIP_LIB=$PWD/alg_amd/libalg_hip_old.so; else unset ALG_HIP_LIB; fi
    python scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 20 2>&1 | grep "TFLOP/s$" >> gpurun_out/epi_ab.log
  done
done
unset ALG_HIP_LIB
cat gpurun_out/epi_ab.log
