#!/bin/bash
# Evidence run (rounds 5 and 6: TAG=r6).  usage: [TAG=r6] bash scripts/gpu_r5_profile.sh [tests] [bench] [prof] [profwl] [pmc]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=${TAG:-r5}
for w in ${@:-bench prof}; do
  case $w in
    tests)
      cd $R
      ALG_PARITY_REPORT=1 timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/${TAG}_pytest_gpu_full.txt 2>&1
      tail -4 $O/${TAG}_pytest_gpu_full.txt ;;
    bench)
      cd $R
      timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_steps20.json 2> $O/${TAG}_bench.err; echo "bench exit $?"
      tail -c 600 $O/${TAG}_bench_driver_steps20.json; tail -3 $O/${TAG}_bench.err ;;
    prof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd5 -o $TAG -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-workloads --no-ab --no-calibration > $O/${TAG}_bench_default_under_rocprof.json 2> $O/profd5.err
      echo "prof exit $?"
      python $R/scripts/rocprof_timed_region.py $(find $O/profd5 -name "*kernel_trace*" | head -1) $O/${TAG}_bench_default_under_rocprof.json > $O/${TAG}_bench_default_rocprof_vs_events.json 2>&1
      cat $O/${TAG}_bench_default_rocprof_vs_events.json
      find $O/profd5 -name "*kernel_trace*" -delete
      cp $(find $O/profd5 -name "*kernel_stats*" | head -1) $O/${TAG}_bench_default_kernel_stats.csv
      rm -rf $O/profd5
      head -8 $O/${TAG}_bench_default_kernel_stats.csv | cut -c1-160 ;;
    profwl)
      cd /tmp
      for wl in c3 c4 c5; do
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd5 -o $TAG -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-ab --no-calibration > $O/${TAG}_bench_${wl}_under_rocprof.json 2> $O/profd5.err
        echo "prof $wl exit $?"
        find $O/profd5 -name "*kernel_trace*" -delete
        cp $(find $O/profd5 -name "*kernel_stats*" | head -1) $O/${TAG}_bench_${wl}_kernel_stats.csv
        rm -rf $O/profd5
        head -4 $O/${TAG}_bench_${wl}_kernel_stats.csv | cut -c1-160
      done ;;
    pmc)
      cd /tmp
      P=$O/pmc_r5; rm -rf $P; mkdir -p $P
      i=0
      for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
                  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
                  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $P/kb_p$i -o p -- python $R/scripts/kbench.py --only attn_model_scores,attn128,gemm_qkv,gemm_out,gemm_ff1,gemm_ff2,ln_mod,qk_norm_rope --iters 2 > /dev/null 2> $P/kb_p$i.err
      done
      python $R/scripts/pmc_summary.py $P > $O/${TAG}_pmc_summary.txt 2>&1
      rm -rf $P
      grep -c mean $O/${TAG}_pmc_summary.txt; tail -12 $O/${TAG}_pmc_summary.txt | cut -c1-200 ;;
  esac
done
