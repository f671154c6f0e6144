#!/bin/bash
# bench.py under different environment switches: each argument is "NAME=VALUE,NAME=VALUE" (or "base")
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/bench_env_ab.log
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  env $envs timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['timed_regions_ms_per_step'], 'relres', d['relres_after_steps'])
r=d['roofline']; print(r['kernel'][:50], r['avg_launch_ms'], r['bytes_per_launch'], r['frac'])
print({k:round(v['avg_ms']*1e3,1) for k,v in r['kernels'].items()})" >> gpurun_out/bench_env_ab.log 2>&1
done
cat gpurun_out/bench_env_ab.log
