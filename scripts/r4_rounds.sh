#!/bin/bash
set -u
out=gpurun_out/r4s
mkdir -p $out
python -m pytest tests/test_four_stage.py -x -q > $out/tests2.log 2>&1 || { tail -30 $out/tests2.log; exit 1; }
tail -2 $out/tests2.log
python bench.py --cells 512 --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/b2_512.json 2> $out/b2_512.err
MG_MARCH4_SEGS=1 python bench.py --cells 512 --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/b2_512_segs1.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s/b2_*.json")):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    k=d["roofline"]["kernels"]; g=d["roofline"].get("inplane_tiles",{})
    print(f.split('/')[-1], d["ms_per_step"], "four", k.get("L1:four-stage",{}).get("avg_ms"), "R1", k["L1:restrict"]["avg_ms"], "tiles", g.get("tiles_per_line"), g.get("tiles_per_column"), g.get("TX"), g.get("TY"), g.get("workgroups"), g.get("schedule"))
PY
