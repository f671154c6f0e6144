#!/bin/bash
# marching restriction: parity, then bench.py restriction time per grid size with the form on / off
set -u
out=gpurun_out/r4mr
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_marchr.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -1 $out/tests.log
for c in ${SIZES:-256 400 512}; do
for v in on off; do
  e=MG_NO_MARCHR=0; [ $v = off ] && e=MG_NO_MARCHR=1
  env MG_DEBUG_FORMAT=1 $e python bench.py --cells $c --steps 10 --warmup 3 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/c${c}_$v.json 2> $out/c${c}_$v.err
  python - $c $v <<'PY'
import json,sys
f=f"c{sys.argv[1]}_{sys.argv[2]}"
g=[l.strip() for l in open(f"gpurun_out/r4mr/{f}.err") if "marchr" in l][:3]
try:
    d=json.loads([l for l in open(f"gpurun_out/r4mr/{f}.json").read().splitlines() if l.startswith('{')][-1])
    k=d["roofline"]["kernels"]
    row=[f"{n} {v['avg_ms']*1e3:.1f}" for n,v in k.items() if "restrict" in n]
    print(f, d["ms_per_step"], "|", ", ".join(row))
except Exception as e:
    print(f, "unreadable", e)
for l in g: print("    ", l[l.index("marchr"):])
PY
done
done
