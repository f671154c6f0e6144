set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_sa_amg.py -q -x -m gpu -k "march or golden or solveMG or cycle_types or graph or sa or rowclass or block_rhs or kcycle" 2>&1 | tail -4 | tee gpurun_out/rs_test.log && rm -f gpurun_out/bench_env_ab.log && bash scripts/bench_env_ab.sh base MG_NO_RESTRICT_SCALE=1
