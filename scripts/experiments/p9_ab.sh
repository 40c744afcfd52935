#!/bin/bash
# kbench A/B of schedule-9 variant libraries (built by p9_build_variant.sh): p9_ab.sh <out> <rounds> <name> [<name> ...]
# "base" = the in-tree library with ALG_GEMM_PIPE=9, "p6" = the in-tree library with the default schedule.
out=$1; rounds=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq $rounds); do
  for v in "$@"; do
    lib=$R/alg_amd/libalg_hip_$v.so; pipe=9
    [ "$v" = base ] && lib=$R/alg_amd/libalg_hip.so
    [ "$v" = p6 ] && lib=$R/alg_amd/libalg_hip.so && pipe=6
    echo "== $v round $r" >> $out
    ALG_HIP_LIB=$lib ALG_GEMM_PIPE=$pipe python $R/scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 12 2>/dev/null | grep "^gemm" >> $out
  done
done
python3 - "$out" <<'PY'
import re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
v = None
for ln in open(sys.argv[1]):
    m = re.match(r"== (\S+) round", ln)
    if m: v = m.group(1); continue
    m = re.match(r"(gemm_\w+)\s+([\d.]+) ms \(best\s+([\d.]+)\)\s+([\d.]+) TFLOP", ln)
    if m: acc[v][m.group(1)].append(float(m.group(4)))
names = ["gemm_qk", "gemm_vt", "gemm_out", "gemm_ff1", "gemm_ff2"]
print("%-10s" % "variant" + "".join("%10s" % n[5:] for n in names))
for v, d in acc.items():
    print("%-10s" % v + "".join("%10.0f" % (sorted(d[n])[len(d[n]) // 2] if d[n] else 0) for n in names))
PY
