#!/bin/bash
# round 4: four-stage pass - parity tests, then the C2 step with each workgroup shape and without it (same box)
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_four_stage.py -x -q -m gpu > gpurun_out/r4/test_four.log 2>&1 || { tail -40 gpurun_out/r4/test_four.log; exit 1; }
tail -3 gpurun_out/r4/test_four.log
for nt in 768 1024 512; do
  MG_MARCH4_NT=$nt MG_DEBUG_FORMAT=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-generic-pass > gpurun_out/r4/bench_nt$nt.json 2> gpurun_out/r4/bench_nt$nt.log
  python - <<PY
import json
d = json.load(open("gpurun_out/r4/bench_nt$nt.json"))
print("nt $nt ms_per_step", d["ms_per_step"], {k: v["avg_ms"] for k, v in d["roofline"]["kernels"].items() if k.startswith("L1")})
PY
done
MG_NO_MARCH4=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-generic-pass > gpurun_out/r4/bench_no4.json 2> gpurun_out/r4/bench_no4.log
python - <<PY
import json
d = json.load(open("gpurun_out/r4/bench_no4.json"))
print("no4 ms_per_step", d["ms_per_step"], {k: v["avg_ms"] for k, v in d["roofline"]["kernels"].items() if k.startswith("L1")})
PY
