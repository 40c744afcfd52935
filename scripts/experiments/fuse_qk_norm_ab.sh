#!/bin/bash
# same-box A/B of QK LayerNorm + rope inside the Q|K store loop (alg_gemm_bf16_pair_qk) against the stand-alone kernel behind the
# pair launch, inside the bench (driver form, 20 steps), arms interleaved twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r4_fuse_qk_norm_ab.txt
for rep in 1 2; do
  for arm in 1 0; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --set fuse_qk_norm=$arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['roofline']['extra']; ts=e['time_share']
fam=sum(v for k,v in ts.items() if k in ('gemm_qkv','qk_norm_rope'))
print('fuse_qk_norm=$arm rep $rep  frames/s %.4f  ms/step %.2f  qkv+norm ms/step %.2f  attn %.0f  gemm_all %.0f' % (d['value'], d['ms_per_step'], fam*d['ms_per_step'], d['roofline']['achieved'], e['gemm_all_tflops']))" | tee -a $O/r4_fuse_qk_norm_ab.txt
  done
done
