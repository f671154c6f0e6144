set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "prolongation_with_staged or gmres_coarse or march2" 2>&1 | tail -8 | tee gpurun_out/winp_test.log
