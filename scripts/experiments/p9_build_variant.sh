#!/bin/bash
# Build libalg_hip_<name>.so with a variant of the schedule-9 main loop: scripts/gen_gemm_p9.py is run with the given
# environment (P9_* knobs), only gemm_p9.o is recompiled (into build_<name>/), everything else is linked from build/.
#   scripts/experiments/p9_build_variant.sh <name> [P9_KNOB=value ...]      (P9_PK=1 in the caller's environment: packed fp32 ops allowed)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/alg_amd/csrc
mkdir -p $cs/build_$name
env "$@" P9_OUT=$cs/build_$name/gemm_p9_loop.inc python $root/scripts/gen_gemm_p9.py
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function \
  $( [ "$P9_PK" = 1 ] || echo "-Xclang -target-feature -Xclang -packed-fp32-ops" ) $P9_CXX -DALG_P9_LOOP_INC="\"$cs/build_$name/gemm_p9_loop.inc\"" -c $cs/gemm_p9.hip -o $cs/build_$name/gemm_p9.o 2>&1 | grep -v "recognized feature" || true
objs=$(ls $cs/build/*.o | grep -v "build/gemm_p9" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $cs/build_$name/gemm_p9.o -o $root/alg_amd/libalg_hip_$name.so
echo built $root/alg_amd/libalg_hip_$name.so
