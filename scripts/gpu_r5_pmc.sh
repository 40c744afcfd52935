#!/bin/bash
# Round-5 counter passes on the attention micro-benchmarks.  usage: bash scripts/gpu_r5_pmc.sh <tag> <kbench case> [ENV=VALUE ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
tag=$1; shift; what=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp
P=$O/pmc_$tag; rm -rf $P; mkdir -p $P
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $P/kb_p$i -o p -- python $R/scripts/kbench.py --only $what --iters 2 > /dev/null 2> $P/kb_p$i.err
done
python $R/scripts/pmc_summary.py $P > $O/r5_pmc_$tag.txt 2>&1
rm -rf $P
cat $O/r5_pmc_$tag.txt | cut -c1-170
