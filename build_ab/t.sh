set -e
timeout -k 10 600 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py tests/test_gpu_ddp.py -q -m gpu -x 2>&1 | tail -3
A_LIB=build_ab/lib_prev.so timeout -k 10 600 profiles/tools/ab.sh whisper-sae_amd/whisper_sae/libwsae_hip.so 2
