#!/bin/bash
# One GPU-box session: GPU tests, smoke, benches, rocprof.  Everything lands in gpurun_out/.
# usage: bash scripts/gpu_session.sh [tests] [smoke] [quick] [bench] [kbench] [gsweep] [pmc] [prof]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
WHAT=${@:-tests smoke quick bench prof}
for w in $WHAT; do
  case $w in
    tests)
      rm -f $O/parity_floor.jsonl
      ALG_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -400 > $O/pytest_gpu.log
      echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest_gpu.log
      tail -25 $O/pytest_gpu.log ;;
    smoke)
      timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -3 $O/smoke.log ;;
    quick)
      timeout 900 python bench.py --layers 4 --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err
      echo "quick exit $?"; tail -c 3000 $O/bench_quick.json; tail -5 $O/bench_quick.err ;;
    driverbench)
      timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
      echo "driverbench exit $?"; tail -c 5000 $O/bench_driver.json; tail -5 $O/bench_driver.err ;;
    bench)
      timeout 1800 python bench.py --cross-check > $O/bench_full.json 2> $O/bench_full.err
      echo "bench exit $?"; tail -c 4000 $O/bench_full.json; tail -5 $O/bench_full.err ;;
    kbench)
      for v in ${ATTN_VARIANTS:-1 13}; do
        echo "== ALG_ATTN_VARIANT=$v" | tee -a $O/kbench.log
        ALG_ATTN_VARIANT=$v timeout 600 python scripts/kbench.py --only attn --check 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.log
      done
      for v in 0; do
        echo "== ALG_GEMM_PIPE=$v" | tee -a $O/kbench.log
        ALG_GEMM_PIPE=$v timeout 600 python scripts/kbench.py --only gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.log
      done ;;
    gsweep)
      for gm in 1 2 4 8 16 32 1000; do
        echo "== ALG_GEMM_GROUP_M=$gm" | tee -a $O/gsweep.log
        ALG_GEMM_GROUP_M=$gm timeout 600 python scripts/kbench.py --only gemm_qk,gemm_ff1,gemm_ff2 2>&1 | grep -v -E "amdgpu.ids|^\{" | tee -a $O/gsweep.log
      done ;;
    pmc)
      cd /tmp
      mkdir -p $O/pmc; rocprofv3 -L > $O/pmc_list.txt 2>&1
      for k in ${PMC_KERNELS:-attn gemm_qk gemm_ff2}; do
        i=0
        for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
                    "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
          i=$((i+1))
          timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc/${k}${PMC_TAG:-}_p$i -o p -- python $R/scripts/kbench.py --only $k --iters 2 > /dev/null 2> $O/pmc/${k}${PMC_TAG:-}_p$i.err
        done
      done
      cd $R
      python scripts/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt | tail -60
      find $O/pmc -name "*.csv" -size +4M -delete ;;
    profdefault)
      # rocprofv3 kernel stats of the DEFAULT bench command (what profiles/ is judged against); the trace is dropped
      cd /tmp
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd -o r1 -- python $R/bench.py > $O/profd_bench.json 2> $O/profd.err
      echo "profdefault exit $?"
      cd $R
      rm -f $O/profd/*kernel_trace*; ls -la $O/profd; tail -c 600 $O/profd_bench.json ;;
    prof)
      cd /tmp
      timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
      echo "prof exit $?"
      cd $R
      find $O/prof -name "*stats*" | head; ls -la $O/prof | head
      # keep only the small summaries (the per-dispatch trace is large)
      find $O/prof -name "*kernel_trace*" -size +8M -delete ;;
  esac
done
