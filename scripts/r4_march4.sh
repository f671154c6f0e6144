#!/bin/bash
# round 4: four-stage pass - parity tests, then the C2 step with each workgroup shape and without it (same box)
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_four_stage.py -q -m gpu > gpurun_out/r4/test_four.log 2>&1 || { tail -40 gpurun_out/r4/test_four.log; exit 1; }
tail -3 gpurun_out/r4/test_four.log
python scripts/march4_ab.py 256 ${M4_VARIANTS:-1024:0 768:0 512:0} > gpurun_out/r4/m4_geo.txt 2>&1 || true
grep -v "^\[" gpurun_out/r4/m4_geo.txt | tail -12
