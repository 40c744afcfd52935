#!/bin/bash
# PMC comparison of the vendor GEMM kernel and ours on the same tensors (instruction mix, LDS, L2, fetch/write sizes)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vpmc; rm -rf $O; mkdir -p $O
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  for who in vendor ours; do for sh in ff2 ff1; do
    timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/${who}_${sh}_p$i -o p -- python $R/scripts/experiments/vendor_pmc_one.py $who $sh 4 > /dev/null 2> $O/${who}_${sh}_p$i.err
  done; done
done
cd $R
python - <<'PY'
import csv, glob, os
from collections import defaultdict
O = "gpurun_out/vpmc"
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
print(open(O + "/summary.txt").read())
PY
