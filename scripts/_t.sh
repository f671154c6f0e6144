set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_distributed.py -q -x -m gpu -k "kcycle_and_jac_gmres or native_sequencer_plugin" 2>&1 | tail -15 | tee gpurun_out/dist_test.log
