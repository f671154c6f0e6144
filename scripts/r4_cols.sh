#!/bin/bash
# column-wise block solve: parity, then C5 with two streams / one stream / the block kernels
set -u
out=gpurun_out/r4c
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_block_columns.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -1 $out/tests.log
python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_cols2.json 2> $out/c5_cols2.err || tail -5 $out/c5_cols2.err
MG_COLUMNS_STREAMS=1 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_cols1.json 2> /dev/null
MG_NO_COLUMNS=1 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_block.json 2> /dev/null
python - <<'PY'
import json
for f in ("c5_cols2","c5_cols1","c5_block"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r4c/{f}.json").read().splitlines() if l.startswith('{')][-1])
        print(f, d["ms_per_step"], round(d["value"]/1e9,2), d["roofline"]["kernel"][:60], d["roofline"]["frac"])
    except Exception as e: print(f, "unreadable", e)
PY
