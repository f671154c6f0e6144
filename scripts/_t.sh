set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "spmatmul or rowclass or march or golden or solveMG or irregular" 2>&1 | tail -4 | tee gpurun_out/lane_test.log && rm -f gpurun_out/bench_env_ab.log && bash scripts/bench_env_ab.sh base MG_NO_LANE_PAD=1 MGVCYCLE_LIB=$PWD/multigrid.jl_amd/csrc/libmgvcycle_pad1.so
