set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "staged_coarse or spmatmul" 2>&1 | tail -3 && rm -f gpurun_out/bench_ab.log && bash scripts/bench_ab.sh base wp256 wp1024
