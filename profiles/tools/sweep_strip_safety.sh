for s in 0.85 0.95 1.0 1.05 1.1 1.2; do
  WSAE_STRIP_SAFETY=$s python bench.py --no-cpu-baseline --steps 100 --warmup 100 --windows 3 > gpurun_out/sw_$s.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/sw_$s.json').read().strip().splitlines()[-1]); print('$s', round(j['ms_per_step']*1e3,1), j['strip_predict'], j['probe_kernel_us'])"
done
