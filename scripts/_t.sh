set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "staged_coarse or spmatmul or march2 or golden or cycle_types" 2>&1 | tail -12 | tee gpurun_out/winr_test.log && rm -f gpurun_out/bench_env_ab.log && bash scripts/bench_env_ab.sh base MG_NO_WINR=1
