#!/bin/bash
# Round-4 evidence run.  usage: bash scripts/gpu_r4_profile.sh [gemmtests] [bench] [prof] [pmc] [workloads]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=${TAG:-r4}
for w in ${@:-gemmtests bench prof}; do
  case $w in
    gemmtests)
      cd $R
      timeout 900 python -m pytest tests/test_gpu_dit_kernels.py tests/test_gpu_fp8.py tests/test_gpu_full_size.py -q --no-header -p no:cacheprovider -k "gemm or schedule9 or race or forward" 2>&1 | tail -5 > $O/${TAG}_gemm_tests.log
      ALG_HIP_LIB=$R/alg_amd/libalg_hip_exp.so timeout 900 python -m pytest tests/test_gpu_dit_kernels.py -q --no-header -p no:cacheprovider -k "gemm or schedule9 or race" 2>&1 | tail -5 >> $O/${TAG}_gemm_tests.log
      cat $O/${TAG}_gemm_tests.log ;;
    bench)
      cd $R
      timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_steps20.json 2> $O/${TAG}_bench.err; echo "bench exit $?"
      tail -c 1500 $O/${TAG}_bench_driver_steps20.json; tail -3 $O/${TAG}_bench.err ;;
    prof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd4 -o $TAG -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/${TAG}_bench_default_under_rocprof.json 2> $O/profd4.err
      echo "prof exit $?"
      find $O/profd4 -name "*kernel_trace*" -delete
      cp $(find $O/profd4 -name "*kernel_stats*" | head -1) $O/${TAG}_bench_default_kernel_stats.csv
      rm -rf $O/profd4
      head -12 $O/${TAG}_bench_default_kernel_stats.csv | cut -c1-160 ;;
    pmc)
      cd /tmp
      P=$O/pmc_r4; rm -rf $P; mkdir -p $P
      i=0
      for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
                  "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
                  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $P/kb_p$i -o p -- python $R/scripts/kbench.py --only attn_model_scores,gemm_qkv,gemm_out,gemm_ff1,gemm_ff2,ln_mod,qk_norm_rope --iters 2 > /dev/null 2> $P/kb_p$i.err
      done
      python $R/scripts/pmc_summary.py $P > $O/${TAG}_pmc_summary.txt 2>&1
      # the same attention launch on N(0, 8^2) scores (the round-3 PMC case): ~25 % of the waves leave the statement early
      rm -rf $P/kb_p[0-9]
      timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $P/kbwide_p1 -o p -- python $R/scripts/kbench.py --only attn_prescaled --iters 2 > /dev/null 2> $P/kbwide.err
      echo "== the attention launch on N(0, 8^2) log2-unit scores (kbench attn_prescaled; round 3's PMC case)" >> $O/${TAG}_pmc_summary.txt
      python $R/scripts/pmc_summary.py $P 2>&1 | grep -A40 "kbwide_p1" >> $O/${TAG}_pmc_summary.txt
      # calibration of SQ_INSTS_VALU on a kernel whose instruction mix is known exactly (scripts/micro/attn_mix.hip)
      if [ -x $R/scripts/micro/attn_mix ]; then
        timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES --output-format csv -d $P/mix_p1 -o p -- $R/scripts/micro/attn_mix > $P/mix.out 2> $P/mix.err
        python - $P/mix_p1 >> $O/${TAG}_pmc_summary.txt <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection*.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== calibration: scripts/micro/attn_mix (k<FILL, NT>: per MFMA FILL 0 = nothing, 1 = 2 exp, 5 = 2 exp + cvt_pk, 6 = 5 + 2 v_add, 7 = 6 + ds_read)")
for k, c in sorted(acc.items()):
    m = sum(c["SQ_INSTS_MFMA"]) / max(len(c["SQ_INSTS_MFMA"]), 1)
    if m > 0:
        print("  %-70s VALU/MFMA %.3f  LDS/MFMA %.3f" % (k, sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]) / m, sum(c["SQ_INSTS_LDS"]) / len(c["SQ_INSTS_LDS"]) / m))
PY
      fi
      rm -rf $P
      grep -c mean $O/${TAG}_pmc_summary.txt ;;
    pmc128)
      # instruction mix of the two d = 128 self-attention kernels at the Wan-480p shape: 64 queries per wave (default) vs 32
      cd /tmp
      P=$O/pmc128_r4; rm -rf $P; mkdir -p $P
      : > $O/${TAG}_pmc128_summary.txt
      for arm in 1 0; do
        i=0
        for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
                    "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
          i=$((i+1))
          ALG_ATTN128_Q64=$arm timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $P/q$arm\_p$i -o p -- python $R/scripts/kbench.py --only attn128 --iters 2 > /dev/null 2> $P/q$arm\_p$i.err
        done
        echo "== ALG_ATTN128_Q64=$arm (1: flash_attn_d128_q64_kernel, 0: flash_attn_d128_pipe_kernel), kbench attn128: N x 40 heads x 32,760 tokens" >> $O/${TAG}_pmc128_summary.txt
        mkdir -p $P/arm$arm; mv $P/q$arm\_p* $P/arm$arm/
        python $R/scripts/pmc_summary.py $P/arm$arm 2>&1 | grep -v "^$" >> $O/${TAG}_pmc128_summary.txt
      done
      rm -rf $P
      tail -30 $O/${TAG}_pmc128_summary.txt | cut -c1-200 ;;
    workloads)
      cd $R
      for wl in c3 c4 c5; do
        timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_workload_$wl.json 2>> $O/${TAG}_bench.err; echo "$wl exit $?"
      done ;;
  esac
done
