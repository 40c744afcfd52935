#!/bin/bash
# Round-3 evidence run: PMC passes over the kernel micro-benchmark at the N = 2 C2 shape (attention as the product calls it,
# the five GEMMs on the default schedule 9), rocprofv3 kernel stats of the default bench command, the driver-form bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
P=$O/pmc_r3; rm -rf $P; mkdir -p $P
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $P/kb_p$i -o p -- python $R/scripts/kbench.py --only attn_prescaled,gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 --iters 2 > /dev/null 2> $P/kb_p$i.err
done
python $R/scripts/pmc_summary.py $P > $O/r3_pmc_summary.txt 2>&1
rm -rf $P/kb_p[0-9]
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd3 -o r3 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/profd3_bench.json 2> $O/profd3.err
echo "profdefault exit $?"
find $O/profd3 -name "*kernel_trace*" -delete
cp $(find $O/profd3 -name "*kernel_stats*" | head -1) $O/r3_bench_default_kernel_stats.csv
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3_bench_driver_steps20.json 2> $O/r3_bench.err; echo "bench exit $?"
tail -c 600 $O/r3_bench_driver_steps20.json; head -12 $O/r3_bench_default_kernel_stats.csv | cut -c1-150; grep -c mean $O/r3_pmc_summary.txt
