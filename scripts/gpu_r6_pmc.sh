#!/bin/bash
# Round-6 counter passes.  --pmc only, FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), no trace domains.
#   1. over the BENCH command itself (short form of the driver's line: 1 warm-up + 4 timed steps = two 3-sample and two 2-sample
#      steps), summarised per (kernel, grid): roofline.traffic of the bench line is then from the run it annotates
#   2. the filter kernels' HBM-side bytes again (r3_pmc_filters_hbm.txt was two rounds old)
#   3. GEMM schedules 10 / 11 next to schedule 9, and the two d = 64 attention statements, in the kernel micro-benchmark
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6pmc; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
B=$O/bench; rm -rf $B; mkdir -p $B
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $B/bench_p$i -o p -- python bench.py --gpus 1 --steps 4 --warmup 1 --no-ab --no-other-workloads --no-cpu-baseline --no-calibration > $B/bench_p$i.json 2> $B/bench_p$i.err)
done
PMC_BY_GRID=1 python $R/scripts/pmc_summary.py $B > $O/r6_pmc_bench_run.txt 2>&1
F=$O/filters; rm -rf $F; mkdir -p $F
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for c in c2x8 c5x8 wanx8g; do
    timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $F/${c}_p$i -o p -- python $R/scripts/filter_one.py $c 3 > /dev/null 2> $F/${c}_p$i.err
  done
done
python $R/scripts/pmc_summary.py $F > $O/r6_pmc_filters_hbm.txt 2>&1
K=$O/kb; rm -rf $K; mkdir -p $K
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  for arm in "p10 ALG_GEMM_PIPE=10 ALG_ATTN_PP=4" "p9m16 ALG_GEMM_PIPE=9 ALG_ATTN_PP=7"; do
    set -- $arm; tag=$1; shift
    env "$@" timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $K/${tag}_p$i -o p -- python $R/scripts/kbench.py --only gemm_qkv,gemm_out,gemm_ff1,gemm_ff2,gemm_out_p11,gemm_ff1_p11,gemm_ff2_p11,attn_model_scores --iters 2 > /dev/null 2> $K/${tag}_p$i.err
  done
done
python $R/scripts/pmc_summary.py $K > $O/r6_pmc_summary.txt 2>&1
rm -rf $B/bench_p*/ $F/*_p*/ $K/*_p*/
ls -la $O; head -60 $O/r6_pmc_bench_run.txt | cut -c1-190
