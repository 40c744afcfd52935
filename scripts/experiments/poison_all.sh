#!/bin/bash
# register / LDS poison test over the asm kernels of the product build, then the shelved 64-query kernel (EXPERIMENTS build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r4_poison_probe.jsonl
for k in d64_pipe d128_pipe gemm9 gemm9_res; do
  timeout 600 python scripts/experiments/q64_poison_probe.py $k 30 2>&1 | tail -1 >> $O/r4_poison_probe.jsonl
done
for arm in 1 2; do
  ALG_HIP_LIB=$R/alg_amd/libalg_hip_exp.so timeout 600 python scripts/experiments/q64_poison_probe.py d128_q64:$arm 30 2>&1 | tail -1 >> $O/r4_poison_probe.jsonl
done
python - <<PY
import json
for line in open("$O/r4_poison_probe.jsonl"):
    try: d = json.loads(line)
    except Exception: print(line.strip()[:300]); continue
    print(d["kernel"], [(a["parts"], a["pattern"], a["bad"]) for a in d["arms"]])
PY
