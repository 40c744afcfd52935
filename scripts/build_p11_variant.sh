#!/bin/bash
# Builds alg_amd/libalg_hip_<tag>.so = the tree's library with GEMM schedule 11's loop generated under the given knobs
# (scripts/gen_gemm_p11.py: P11_DMA_ROWS, P11_B_ROWS; P11_NO_DMA / P11_NO_READS / P11_NO_B / P11_NO_B_WAIT are timing-only ablations).
# usage: bash scripts/build_p11_variant.sh <tag> [KNOB=VALUE ...]      then: ALG_HIP_LIB=alg_amd/libalg_hip_<tag>.so python scripts/kbench.py ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); tag=$1; shift
D=$R/alg_amd/csrc/build_$tag; mkdir -p $D
TAPFLAG=""; for kv in "$@"; do [ "$kv" = "TAP=1" ] && TAPFLAG="$TAPFLAG -DALG_GEMM_TAP"; [ "$kv" = "NOSTORE=1" ] && TAPFLAG="$TAPFLAG -DALG_ABL_NOSTORE"; case "$kv" in GROUP_M=*) TAPFLAG="$TAPFLAG -DALG_GROUP_M_OVERRIDE=${kv#GROUP_M=}";; esac; done   # TAP=1: the per-workgroup clock tap (scripts/probes/gemm_tap.py)
env "$@" P11_OUT=$D/gemm_p11_loop.inc python $R/scripts/gen_gemm_p11.py > /dev/null
cd $R/alg_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
  $TAPFLAG -DALG_P11_LOOP_INC="\"$D/gemm_p11_loop.inc\"" -c gemm_p11.hip -o $D/gemm_p11.o
objs=$(ls build/*.o | grep -v "build/gemm_p11.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $D/gemm_p11.o -o ../libalg_hip_$tag.so
echo built alg_amd/libalg_hip_$tag.so
