#!/bin/bash
# bench.py on each experiment build named on the command line ("" = the shipped library)
mkdir -p gpurun_out
for n in "$@"; do
  [ "$n" = base ] && n=""
  lib=multigrid.jl_amd/csrc/libmgvcycle${n:+_}$n.so
  echo "== $lib" >> gpurun_out/bench_ab.log
  MGVCYCLE_LIB=$PWD/$lib timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['timed_regions_ms_per_step'])
print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})" >> gpurun_out/bench_ab.log 2>&1
done
cat gpurun_out/bench_ab.log
