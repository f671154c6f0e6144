set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -q -x -m gpu -k "rowclass or march or tile or box or golden or gmres_coarse or graph" 2>&1 | tail -6 | tee gpurun_out/tile_test.log && rm -f gpurun_out/bench_env_ab.log && bash scripts/bench_env_ab.sh base MG_NO_TILE_LANE=1
