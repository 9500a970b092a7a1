#!/bin/bash
# PMC passes for one kernel of the step (default: decode_fast).  Usage: pmc_decode.sh [kernel-regex]
# Each counter group is its own rocprofv3 run (the hardware cannot collect them together).
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-decode_fast}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "$K" --kernel-trace --output-format csv \
      -d $R/gpurun_out/pmc_k$i -- python3 $R/bench.py --steps 4 --warmup 2 --ring-rows 262144 --no-cpu-baseline \
      > $R/gpurun_out/pmc_k$i.log 2>&1
  echo "group $i rc=$?"
done
