#!/bin/bash
set -u
out=gpurun_out/r4m27
mkdir -p $out
for c in 400 512; do
for v in on off; do
  e=MG_NO_MARCH27=0; [ $v = off ] && e=MG_NO_MARCH27=1
  env MG_DEBUG_FORMAT=1 $e python bench.py --cells $c --steps 10 --warmup 3 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/c${c}_$v.json 2> $out/c${c}_$v.err
  python - $c $v <<'PY'
import json,sys
f=f"c{sys.argv[1]}_{sys.argv[2]}"
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
done
done
