#!/bin/bash
# same-box A/B of the paired Q|K + V^T launch inside the bench (driver form, 20 steps), arms interleaved twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r4_pair_qkv_ab.txt
for rep in 1 2; do
  for arm in 1 0; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --set pair_qkv=$arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['roofline']['extra']; ts=e['time_share']
fam=sum(v for k,v in ts.items() if k in ('gemm_qk','gemm_vt','gemm_qkv'))
print('pair_qkv=$arm rep $rep  frames/s %.4f  ms/step %.2f  qk+vt ms/step %.2f  attn %.0f  gemm_all %.0f' % (d['value'], d['ms_per_step'], fam*d['ms_per_step'], d['roofline']['achieved'], e['gemm_all_tflops']))" | tee -a $O/r4_pair_qkv_ab.txt
  done
done
