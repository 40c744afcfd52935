#!/bin/bash
# Builds alg_amd/libalg_hip_<tag>.so = the tree's library with GEMM schedule 10's loop generated under the given knobs
# (scripts/gen_gemm_p10.py: P10_B1_ROWS, P10_DMA_GAP, P10_B_EARLY_GAP; P10_NO_DMA / P10_NO_READS are timing-only ablations).
# usage: bash scripts/build_p10_variant.sh <tag> [KNOB=VALUE ...]      then: ALG_HIP_LIB=alg_amd/libalg_hip_<tag>.so python scripts/kbench.py ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); tag=$1; shift
D=$R/alg_amd/csrc/build_$tag; mkdir -p $D
TAPFLAG=""; for kv in "$@"; do [ "$kv" = "TAP=1" ] && TAPFLAG="$TAPFLAG -DALG_GEMM_TAP"; [ "$kv" = "NOSTORE=1" ] && TAPFLAG="$TAPFLAG -DALG_ABL_NOSTORE"; case "$kv" in GROUP_M=*) TAPFLAG="$TAPFLAG -DALG_GROUP_M_OVERRIDE=${kv#GROUP_M=}";; esac; done   # TAP=1: the per-workgroup clock tap (scripts/probes/gemm_tap.py)
env "$@" P10_OUT=$D/gemm_p10_loop.inc python $R/scripts/gen_gemm_p10.py > /dev/null
cd $R/alg_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
  $TAPFLAG -DALG_P10_LOOP_INC="\"$D/gemm_p10_loop.inc\"" -c gemm_p10.hip -o $D/gemm_p10.o
objs=$(ls build/*.o | grep -v "build/gemm_p10.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $D/gemm_p10.o -o ../libalg_hip_$tag.so
echo built alg_amd/libalg_hip_$tag.so
