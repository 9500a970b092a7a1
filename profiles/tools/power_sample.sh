#!/bin/bash
# Sample the GPU's power draw and clocks (read-only rocm-smi queries) while bench.py runs a long window.
# Usage: bash profiles/tools/power_sample.sh [extra bench.py args]   -> gpurun_out/power_sample.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/power_sample.txt
mkdir -p $R/gpurun_out
: > $OUT
python3 $R/bench.py --steps 30000 --warmup 200 --windows 1 --no-cpu-baseline "$@" > $R/gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 6   # import torch, build the ring
for i in $(seq 1 14); do
  echo "--- sample $i" >> $OUT
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk" >> $OUT
  sleep 0.5
done
wait $BP
cut -c1-200 $R/gpurun_out/power_bench.json >> $OUT
echo "--- idle" >> $OUT
sleep 2
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" >> $OUT
