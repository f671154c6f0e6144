#!/bin/bash
# A/B of the fused sweep + residual pass over tile geometries + PMC traffic of three of them.  Output: gpurun_out/$1/
set -u
out=gpurun_out/${1:-m3}
mkdir -p $out
export TMPDIR=/tmp
python3 scripts/march3_ab.py 256 > $out/ab.txt 2> $out/ab.err
cat $out/ab.txt
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$lc -- python3 scripts/march3_ab.py 256 ${PMC_VARIANTS:-m2 3:2:2 3:2:4} > $out/pmc_$lc.log 2>&1
  f=$(find $out/pmc_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f | grep -E "march2|march3" > $out/pmc_$lc.txt
done
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
cat $out/pmc_fetch_size.txt $out/pmc_write_size.txt
