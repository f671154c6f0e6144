#!/bin/bash
# The generic-format legs of scripts/round3_evidence.sh alone (after a change to the band form): the C2 bench line, kernel
# stats and PMC passes with row classes off.  usage (through gpurun): bash scripts/round3_evidence_generic.sh r03b
set -u
out=gpurun_out/${1:-r03}
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
export MG_NO_ROWCLASS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2g -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass > $out/prof_c2g.log 2>&1
f=$(find $out/prof_c2g -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_generic_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_256g_$lc -- python3 scripts/pmc_probe.py 256 > $out/pmc_256g_$lc.log 2>&1
  f=$(find $out/pmc_256g_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_256_generic_$lc.txt
done
unset MG_NO_ROWCLASS
find $out -name "*.csv" -size +1M -delete
find $out -type d -name "prof_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
grep "march3" $out/c2_generic_kernel_stats.csv | cut -c1-180
grep march3 $out/pmc_256_generic_*.txt | cut -c1-220
