# HBM-side byte counters of the filter kernels (VERDICT r2 missing 2 / next 7): FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC slots: 3 + 2), --pmc only (no trace domains), the three batched shapes.  FETCH_SIZE is in KiB of 64-byte-tallied
# requests: per the guide's HBM section a wide coalesced stream reads 2 x FETCH_SIZE bytes on gfx950; WRITE_SIZE is taken as
# reported (KiB) and calibrated against the kernel's exact output bytes.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcf_hbm; mkdir -p $O
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for c in c2x8 c5x8 wanx8g; do
    timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/${c}_p$i -o p -- python $R/scripts/filter_one.py $c 3 > /dev/null 2> $O/${c}_p$i.err
  done
done
cd $R; python scripts/pmc_summary.py $O > gpurun_out/r3_pmc_filters_hbm.txt 2>&1; cat gpurun_out/r3_pmc_filters_hbm.txt | head -80
