#!/bin/bash
# fixed cost against cost per plane of the 27-point marching form: same tiles (3 x 12 of 43 x 11, 512 threads, one workgroup per CU), 2 / 4 / 7 segments
set -u
out=gpurun_out/r4m27
mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env MG_DEBUG_FORMAT=1 "$@" python bench.py --cells 256 --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/$name.json 2> $out/$name.err
  python - $name <<'PY'
import json,sys
f=sys.argv[1]
g=[l.strip() for l in open(f"gpurun_out/r4m27/{f}.err") if "march27" in l][:2]
try:
    d=json.loads([l for l in open(f"gpurun_out/r4m27/{f}.json").read().splitlines() if l.startswith('{')][-1])
    k=d["roofline"]["kernels"]
    row=[f"{n.split(':')[1]} {v['avg_ms']*1e3:.1f}x{v['launches_per_step']:.0f}" for n,v in k.items() if n.startswith("L2:") and "smooth" in n or n=="L2:residual"]
    print(f, d["ms_per_step"], "|", ", ".join(row))
except Exception as e:
    print(f, "unreadable", e)
for l in g: print("    ", l[l.index("march27"):l.index("LDS")] if "LDS" in l else l)
PY
}
for s in 2 4 7; do run seg$s MG_MARCH27_NT=512 MG_MARCH27_WGS=1 MG_MARCH27_TILES_X=3 MG_MARCH27_SEGS=$s; done
