#!/bin/bash
# the coarse sub-cycle's HIP graph from level 2 instead of level 3 (MG_GRAPH_MAX_ROWS), alternating
set -u
out=gpurun_out/r4g
mkdir -p $out
for i in 1 2; do
for g in 300000 3000000; do
  MG_GRAPH_MAX_ROWS=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/g${g}_$i.json 2> /dev/null
  python - $g $i <<'PY'
import json,sys
f=f"g{sys.argv[1]}_{sys.argv[2]}"
d=json.loads([l for l in open(f"gpurun_out/r4g/{f}.json").read().splitlines() if l.startswith('{')][-1])
print(f, d["ms_per_step"], d.get("regions_ms"))
PY
done
done
