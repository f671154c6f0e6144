set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -q -x -m gpu -k "tiles_of_256 or rowclass or march or golden or box or tiled" 2>&1 | tail -12 | tee gpurun_out/t256_test.log && rm -f gpurun_out/bench_env_ab.log && bash scripts/bench_env_ab.sh base MG_NO_TILE_SMALL=1
