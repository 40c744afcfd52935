cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcf; mkdir -p $O
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for c in c2x8 c5x8 wanx8g; do
    timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/${c}_p$i -o p -- python $R/scripts/filter_one.py $c 3 > /dev/null 2> $O/${c}_p$i.err
  done
done
cd $R; python scripts/pmc_summary.py $O > gpurun_out/pmc_filter_summary.txt 2>&1; grep -i "down_up\|gaussian" gpurun_out/pmc_filter_summary.txt | head -60
