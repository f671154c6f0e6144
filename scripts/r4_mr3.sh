#!/bin/bash
# marching restriction: 32 x 8 (default) against 24 x 10 coarse tiles at 400^3 / 512^3
set -u
out=gpurun_out/r4mr
mkdir -p $out
for c in 400 512; do
for t in "32 8" "24 10"; do
  set -- $t
  MG_MARCHR_TX=$1 MG_MARCHR_TY=$2 python bench.py --cells $c --steps 10 --warmup 3 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/g${c}_$1.json 2> /dev/null
  python - $c $1 <<'PY'
import json,sys
f=f"g{sys.argv[1]}_{sys.argv[2]}"
d=json.loads([l for l in open(f"gpurun_out/r4mr/{f}.json").read().splitlines() if l.startswith('{')][-1])
k=d["roofline"]["kernels"]
print(f, d["ms_per_step"], "L1:restrict", round(k["L1:restrict"]["avg_ms"]*1e3,1), "L2:restrict", round(k["L2:restrict"]["avg_ms"]*1e3,1))
PY
done
done
