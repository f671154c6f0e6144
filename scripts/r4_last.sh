#!/bin/bash
set -u
mkdir -p gpurun_out/r4l
timeout -k 10 600 python -m pytest tests/test_four_stage.py tests/test_block_columns.py tests/test_gpu_parity.py -x -q -k "four_stage or block or solve or golden or cycle" > gpurun_out/r4l/tests.log 2>&1 || { tail -30 gpurun_out/r4l/tests.log; exit 1; }
tail -1 gpurun_out/r4l/tests.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
