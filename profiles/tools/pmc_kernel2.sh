#!/bin/bash
# PMC passes for one kernel of the train step.  Usage: pmc_kernel.sh <kernel-regex> [extra bench.py args]
# Each counter group is its own rocprofv3 run (the hardware cannot collect them together); the per-counter averages
# per dispatch of the matching kernel are printed at the end (gpurun_out/pmc_<regex>.txt).
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-decode_mfma}
shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_k$i
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "$K" --kernel-trace --output-format csv \
      -d $R/gpurun_out/pmc_k$i -- python3 $R/bench.py --steps 4 --warmup 2 --ring-rows 262144 --no-cpu-baseline "$@" \
      > $R/gpurun_out/pmc_k$i.log 2>&1 || { echo "group $i failed"; tail -3 $R/gpurun_out/pmc_k$i.log; exit 1; }
done
python3 - "$R" "$K" <<'PY' | tee $R/gpurun_out/pmc_${K}.txt
import csv, glob, sys, collections
root, k = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{root}/gpurun_out/pmc_k*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"kernel /{k}/: average per dispatch")
for name in sorted(acc):
    v = acc[name]
    print(f"  {name:32s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
