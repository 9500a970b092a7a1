#!/bin/bash
# The three rocprofv3 passes behind profiles/rNN_*: kernel-trace stats of the default bench command, then FETCH_SIZE
# and WRITE_SIZE in their own PMC passes (one counter per pass; never combined with a sys/hip trace).
# Usage: bash profiles/tools/profile_step.sh <tag> [extra bench.py args]     (run on the GPU box)
# Output: gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.csv, gpurun_out/<tag>_pmc_traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
shift
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
    python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $OUT/${TAG}_stats.log 2>&1 \
    || { echo "stats pass failed"; tail -5 $OUT/${TAG}_stats.log; exit 1; }
tail -1 $OUT/${TAG}_stats.log > $OUT/${TAG}_bench_under_rocprof.json
f=$(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/${TAG}_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python3 $R/bench.py --steps 5 --warmup 2 --ring-rows 262144 --no-cpu-baseline "$@" > $d.log 2>&1 \
      || { echo "$c pass failed"; tail -5 $d.log; exit 1; }
done
python3 $R/profiles/tools/summarize_traffic.py $OUT $TAG
