#!/bin/bash
set -u
out=gpurun_out/r4m27
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace27 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad > $out/trace27.log 2>&1
t=$(find $out/trace27 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 profiles/summarize_trace.py $t > $out/trace27_by_grid.md
grep -E "march27|tile_spmv|kernel \|" $out/trace27_by_grid.md | head -12
rm -rf $out/trace27
