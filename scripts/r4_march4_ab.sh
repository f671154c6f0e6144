#!/bin/bash
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_four_stage.py -q -m gpu > gpurun_out/r4/test_four.log 2>&1 || { tail -40 gpurun_out/r4/test_four.log; exit 1; }
tail -2 gpurun_out/r4/test_four.log
python scripts/march4_ab.py 256 ${M4_VARIANTS:-1024:0 1024:4 1024:5 1024:7 768:0 768:5 512:0} > gpurun_out/r4/m4_geo.txt 2>&1 || true
grep -v "^\[" gpurun_out/r4/m4_geo.txt | grep four-stage
