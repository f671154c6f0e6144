#!/bin/bash
# A/B of the L2-tiled block order of the single-vector lane kernel (restriction) and the four-stage geometry at 400^3 / 512^3
set -u
out=gpurun_out/r4s
mkdir -p $out
python -m pytest tests/test_four_stage.py -x -q > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for c in 256 400; do
  python bench.py --cells $c --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/b_${c}.json 2> $out/b_${c}.err
  MG_NO_SCHED_LANE=1 python bench.py --cells $c --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/b_${c}_nosched.json 2> /dev/null
  echo "$c done"
done
python bench.py --cells 512 --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/b_512.json 2> $out/b_512.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s/b_*.json")):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k=d["roofline"]["kernels"]
    g=d["roofline"].get("inplane_tiles",{})
    print(f.split('/')[-1], d["ms_per_step"], "four", k.get("L1:four-stage",{}).get("avg_ms"), "R1", k["L1:restrict"]["avg_ms"], "P1", k["L1:prolong"]["avg_ms"], "R2", k.get("L2:restrict",{}).get("avg_ms"), "tiles", g.get("tiles_per_line"), g.get("tiles_per_column"), g.get("TX"), g.get("TY"), g.get("schedule"))
PY
