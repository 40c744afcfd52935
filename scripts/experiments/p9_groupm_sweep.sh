#!/bin/bash
# schedule 9: tiles per M group of the XCD-aware tile order (ALG_GEMM_GROUP_M), interleaved rounds, kbench medians
out=$1; rounds=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -f $out
for r in $(seq $rounds); do
  for g in "$@"; do
    echo "== g$g round $r" >> $out
    ALG_GEMM_GROUP_M=$g python $R/scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 12 2>/dev/null | grep "^gemm" >> $out
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
    if m: acc[v][m.group(1)].append(float(m.group(2)))
names = ["gemm_qk", "gemm_vt", "gemm_out", "gemm_ff1", "gemm_ff2"]
print("%-8s" % "GROUP_M" + "".join("%10s" % n[5:] for n in names) + "   sum ms")
for v, d in acc.items():
    med = [sorted(d[n])[len(d[n]) // 2] for n in names]
    print("%-8s" % v + "".join("%10.3f" % m for m in med) + "   %.3f" % sum(med))
PY
