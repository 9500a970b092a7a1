set -e
WSAE_LIB=$PWD/build_ab/lib_s16.so timeout -k 10 600 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -15
timeout -k 10 600 profiles/tools/ab.sh build_ab/lib_s16.so 2
