set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_krylov.py tests/test_gpu_parity.py -q -x -m gpu -k "krylov or wrapper or nonzero or solveMG or block" 2>&1 | tail -4 | tee gpurun_out/host_test.log && python scripts/diag_host_api.py 2>&1 | grep "host API" | tee gpurun_out/host_api.txt
