set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -x -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/gpu_suite.log
