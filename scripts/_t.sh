set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_krylov.py tests/test_gpu_parity.py -q -x -m gpu -k "krylov or pcg or cg or wrapper or march" 2>&1 | tail -5 | tee gpurun_out/pcg_test.log && (python scripts/diag_pcg.py 2>&1 | grep -v Warn | grep -v amdgpu) | tee gpurun_out/diag_pcg.txt
