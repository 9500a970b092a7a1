#!/bin/bash
# Same-box A/B of two library builds on the ReLU step (bench.py --relu), alternating runs.
#   A_LIB=build_ab/libwsae_a.so bash profiles/tools/ab_relu.sh <B lib> [rounds] [extra bench args]
B_LIB=$1; ROUNDS=${2:-2}; shift 2
A_LIB=${A_LIB:-whisper-sae_amd/whisper_sae/libwsae_hip.so}
for r in $(seq $ROUNDS); do
  for tag in A B; do
    lib=$A_LIB; [ $tag = B ] && lib=$B_LIB
    WSAE_LIB=$(realpath $lib) python3 bench.py --relu --no-cpu-baseline --steps 100 --warmup 30 --windows 3 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$tag', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M act/s')"
  done
done
