#!/bin/bash
# marching restriction: geometry sweep on C2 (coarse tile, segments)
set -u
out=gpurun_out/r4mr
mkdir -p $out
run() {
  name=$1; shift
  env MG_DEBUG_FORMAT=1 "$@" python bench.py --cells 256 --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/$name.json 2> $out/$name.err
  python - $name <<'PY'
import json,sys
f=sys.argv[1]
g=[l.strip() for l in open(f"gpurun_out/r4mr/{f}.err") if "marchr" in l][:1]
d=json.loads([l for l in open(f"gpurun_out/r4mr/{f}.json").read().splitlines() if l.startswith('{')][-1])
k=d["roofline"]["kernels"]
print(f, d["ms_per_step"], "| L1:restrict", round(k["L1:restrict"]["avg_ms"]*1e3,1), "|", g[0][g[0].index("tiles"):] if g else "")
PY
}
run d
run s6 MG_MARCHR_SEGS=6
run s8 MG_MARCHR_SEGS=8
run s16 MG_MARCHR_SEGS=16
run s24 MG_MARCHR_SEGS=24
run t16 MG_MARCHR_TX=16 MG_MARCHR_TY=16
run t16s8 MG_MARCHR_TX=16 MG_MARCHR_TY=16 MG_MARCHR_SEGS=8
run t32x4 MG_MARCHR_TX=32 MG_MARCHR_TY=4
run t24 MG_MARCHR_TX=24 MG_MARCHR_TY=10
