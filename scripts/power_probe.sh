#!/bin/bash
# Sample GPU power / clocks while a kernel loop runs (is the chip power-limited under our kernels?).
# usage: bash scripts/power_probe.sh <tag> <command...>
tag=$1; shift
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/power_$tag.log
mkdir -p $(dirname $out)
( "$@" > ${out%.log}.cmd.log 2>&1 ) &
pid=$!
sleep 2
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|fclk" | tr '\n' ' ' >> $out
  echo >> $out
  kill -0 $pid 2>/dev/null || break
  sleep 0.5
done
wait $pid
tail -4 $out
