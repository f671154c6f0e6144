set -o pipefail
mkdir -p gpurun_out
MGVCYCLE_LIB=$PWD/multigrid.jl_amd/csrc/libmgvcycle.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "march2" 2>&1 | tail -5 | tee gpurun_out/march2_test.log && timeout -k 10 800 python scripts/march2_ab.py 2>&1 | grep -v Warning | tee gpurun_out/march2_ab.log
