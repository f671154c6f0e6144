set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_krylov.py -q -x -m gpu -k "march2 or pcg or golden or nonzero_initial or krylov or solveMG" 2>&1 | tail -5 | tee gpurun_out/zero_test.log && (python scripts/diag_pcg.py 2>&1 | grep -v Warn; MG_NO_MARCH2_ZERO=1 python scripts/diag_pcg.py 2>&1 | grep -v Warn; MG_NO_MARCH2=1 MG_NO_TILE_LANE=1 MG_NO_WINP=1 python scripts/diag_pcg.py 2>&1 | grep -v Warn) | tee gpurun_out/diag_pcg.txt
