#!/bin/bash
# same-box A/B/C of the whole step: old library, current build (operands hoisted everywhere), residual kernels with the old loads
mkdir -p gpurun_out; rm -f gpurun_out/epi_bench_ab3.log
for rep in 1 2; do
  for lib in old new reso; do
    case $lib in old) export ALG_HIP_LIB=$PWD/alg_amd/libalg_hip_old.so;; reso) export ALG_HIP_LIB=$PWD/alg_amd/libalg_hip_reso.so;; *) unset ALG_HIP_LIB;; esac
    python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline']['extra']
print('$lib rep $rep', round(d['value'],4), 'f/s', round(d['ms_per_step'],1), 'ms/step attn', round(d['roofline']['achieved']), 'gemm', {k[5:-7]:round(v) for k,v in e.items() if k.startswith('gemm_') and k.endswith('_tflops')})" >> gpurun_out/epi_bench_ab3.log
  done
done
unset ALG_HIP_LIB
cat gpurun_out/epi_bench_ab3.log
