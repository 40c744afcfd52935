#!/bin/bash
# round 3: schedule 9 next to the vendor kernel -- scoreboard on the plain shapes, power / clocks while looping, PMC counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/p9v; rm -rf $O; mkdir -p $O
cd $R
for p in 6 9; do echo "== ALG_GEMM_PIPE=$p" >> $O/scoreboard.txt; ALG_GEMM_PIPE=$p python scripts/vendor_gemm_probe.py 2>/dev/null >> $O/scoreboard.txt; done
for sh in ff1 ff2; do
  bash scripts/power_probe.sh vendor_$sh python scripts/experiments/gemm_loop_one.py vendor $sh 6 > /dev/null
  ALG_GEMM_PIPE=9 bash scripts/power_probe.sh p9_$sh python scripts/experiments/gemm_loop_one.py ours $sh 6 > /dev/null
  ALG_GEMM_PIPE=6 bash scripts/power_probe.sh p6_$sh python scripts/experiments/gemm_loop_one.py ours $sh 6 > /dev/null
done
cat gpurun_out/power_*.log gpurun_out/power_*.cmd.log > $O/power.txt 2>/dev/null
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  for sh in ff1 ff2; do
    ALG_GEMM_PIPE=9 timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/ours9_${sh}_p$i -o p -- python $R/scripts/experiments/vendor_pmc_one.py ours $sh 4 > /dev/null 2> $O/ours9_${sh}_p$i.err
    [ $i -le 3 ] || [ $i -eq 5 ] && timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/vendor_${sh}_p$i -o p -- python $R/scripts/experiments/vendor_pmc_one.py vendor $sh 4 > /dev/null 2> $O/vendor_${sh}_p$i.err
  done
done
cd $R
python - <<'PY'
import csv, glob, os
from collections import defaultdict
O = "gpurun_out/p9v"
res = defaultdict(dict)
for d in sorted(glob.glob(O + "/*_p*")):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d).rsplit("_p", 1)[0]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "gemm" not in k.lower() and "Cijk" not in k: continue
            acc[k[:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[(tag, k)][c] = sum(v) / len(v)
with open(O + "/summary.txt", "w") as out:
    for (tag, k), cs in sorted(res.items()):
        out.write("== %s  %s\n" % (tag, k))
        for c, v in sorted(cs.items()):
            out.write("   %-32s %.4g\n" % (c, v))
PY
rm -rf $O/*_p[0-9]
cat $O/scoreboard.txt; cat $O/power.txt
