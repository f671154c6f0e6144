#!/bin/bash
# A/B of experiment builds (make variant NAME=.. DEFS=..): march-kernel parity tests on each library, then timing.
set -o pipefail
mkdir -p gpurun_out
CS=multigrid.jl_amd/csrc
for lib in $CS/libmgvcycle*.so; do
  echo "== parity $lib" | tee -a gpurun_out/ab.log
  MGVCYCLE_LIB=$PWD/$lib timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "march or rowclass_variants_on_odd_grids or rowclass_exception_rows" 2>&1 | tail -3 | tee -a gpurun_out/ab.log || exit 1
done
timeout -k 10 900 python scripts/march_ab.py 2>&1 | grep -v Warning | tee -a gpurun_out/ab.log
