#!/bin/bash
# 27-point marching form: parity, then time per launch against threads per workgroup / workgroups per CU / planes per segment
set -u
out=gpurun_out/r4m27
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_march27.py -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log
run() {  # name, env...
  name=$1; shift
  env MG_DEBUG_FORMAT=1 "$@" python bench.py --cells ${CELLS:-256} --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/$name.json 2> $out/$name.err
  python - $name <<'PY'
import json,sys
f=sys.argv[1]
g=[l.strip() for l in open(f"gpurun_out/r4m27/{f}.err") if "march27" in l][:4]
try:
    d=json.loads([l for l in open(f"gpurun_out/r4m27/{f}.json").read().splitlines() if l.startswith('{')][-1])
    k=d["roofline"]["kernels"]
    row=[f"{n} {v['avg_ms']*1e3:.1f}x{v['launches_per_step']:.0f}" for n,v in k.items() if (n.startswith("L2:") or n.startswith("L3:")) and ("smooth" in n or "residual" in n)]
    print(f, d["ms_per_step"], "|", ", ".join(row))
except Exception as e:
    print(f, "unreadable", e)
for l in g: print("    ", l[l.index("march27"):l.index("LDS")] if "LDS" in l else l)
PY
}
run off MG_NO_MARCH27=1
run auto
run n1024 MG_MARCH27_NT=1024
run n512w1 MG_MARCH27_NT=512 MG_MARCH27_WGS=1
run n512w2 MG_MARCH27_NT=512 MG_MARCH27_WGS=2
run n256w4 MG_MARCH27_NT=256 MG_MARCH27_WGS=4
run l3 MG_MARCH27_MIN_ROWS=100000
