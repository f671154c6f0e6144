#!/bin/bash
# Round-3 evidence on one MI355X box: bench lines, rocprofv3 kernel stats, PMC passes.  Output: gpurun_out/$1/
# usage (through gpurun): bash scripts/round3_evidence.sh r03a
set -u
out=gpurun_out/${1:-r03}
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
echo "c2 done" > $out/progress.txt
python bench.py --cells 400 --steps 20 --warmup 5 --no-cpu-baseline > $out/c2_400_bench.json 2> $out/c2_400_bench.err
echo "c2-400 done" >> $out/progress.txt
python bench.py --force-sharded-path --no-cpu-baseline --steps 20 --warmup 5 > $out/c2_sharded_w1_bench.json 2> $out/c2_sharded_w1.err
python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_bench.json 2> $out/c5_bench.err
echo "sharded, c5 done" >> $out/progress.txt
MG_NO_MARCH3=1 MG_NO_DEAD_T=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/c2_bench_round2_kernels.json 2> $out/c2_bench_round2_kernels.err
# rocprofv3: the program itself after "--"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/prof_c2.log 2>&1
f=$(find $out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
t=$(find $out/prof_c2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 profiles/summarize_trace.py $t > $out/c2_kernel_by_grid.md
echo "kernel trace done" >> $out/progress.txt
# the band form (grid operator without row classes): the whole bench on the generic formats, kernel stats + PMC of its pass
export MG_NO_ROWCLASS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2g -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass > $out/prof_c2g.log 2>&1
f=$(find $out/prof_c2g -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_generic_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_256g_$lc -- python3 scripts/pmc_probe.py 256 > $out/pmc_256g_$lc.log 2>&1
  f=$(find $out/pmc_256g_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_256_generic_$lc.txt
done
unset MG_NO_ROWCLASS
echo "generic done" >> $out/progress.txt
for cells in 256 400; do
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${cells}_$lc -- python3 scripts/pmc_probe.py $cells > $out/pmc_${cells}_$lc.log 2>&1
  f=$(find $out/pmc_${cells}_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_${cells}_$lc.txt
  echo "pmc $cells $c done" >> $out/progress.txt
done
done
find $out -name "*.csv" -size +1M -delete
find $out -type d -name "prof_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
ls -la $out
