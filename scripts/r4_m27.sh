#!/bin/bash
# A/B of the 27-point marching form on C2 (level 2 = 129^3 nodes): parity tests, then bench.py kernel table per variant
set -u
out=gpurun_out/r4m27
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_march27.py -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --cells ${CELLS:-256} --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/$name.json 2> $out/$name.err
  grep "march27" $out/$name.err | head -4
}
run off MG_NO_MARCH27=1
run auto MG_DEBUG_FORMAT=1
run nt512 MG_DEBUG_FORMAT=1 MG_MARCH27_NT=512
run nt1024 MG_DEBUG_FORMAT=1 MG_MARCH27_NT=1024
run l3 MG_DEBUG_FORMAT=1 MG_MARCH27_MIN_ROWS=100000
python - <<'PY'
import json,glob
for f in ("off","auto","nt512","nt1024","l3"):
    try: d=json.loads([l for l in open(f"gpurun_out/r4m27/{f}.json").read().splitlines() if l.startswith('{')][-1])
    except Exception as e: print(f,"unreadable",e); continue
    k=d["roofline"]["kernels"]
    row=[f"{n.split(':')[1]} {v['avg_ms']*1e3:.1f}x{v['launches_per_step']:.0f}" for n,v in k.items() if n.startswith("L2:")]
    row3=[f"{n.split(':')[1]} {v['avg_ms']*1e3:.1f}x{v['launches_per_step']:.0f}" for n,v in k.items() if n.startswith("L3:")]
    print(f, d["ms_per_step"], "| L2:", ", ".join(row), "| L3:", ", ".join(row3))
PY
