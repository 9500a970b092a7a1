set -e
timeout -k 10 600 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 100 --warmup 100 --windows 2 --profile-all 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({k.split('<')[0]:round(v*1e3,1) for k,v in j['kernel_ms_per_step'].items()})"
