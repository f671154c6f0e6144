#!/bin/bash
# column-wise block solve on C5: number of streams
set -u
out=gpurun_out/r4c
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_block_columns.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -1 $out/tests.log
for s in 2 3 4 6 8; do
  MG_COLUMNS_STREAMS=$s python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_s$s.json 2> /dev/null
  python - $s <<'PY'
import json,sys
f=f"c5_s{sys.argv[1]}"
d=json.loads([l for l in open(f"gpurun_out/r4c/{f}.json").read().splitlines() if l.startswith('{')][-1])
print(f, d["ms_per_step"], round(d["value"]/1e9,2))
PY
done
