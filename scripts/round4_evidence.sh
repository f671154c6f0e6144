#!/bin/bash
# Round-4 evidence on one MI355X box.  Output: gpurun_out/$1/   usage (through gpurun): bash scripts/round4_evidence.sh r04a [part]
# part 1: bench lines (C2, A/B switches, 400^3, C5, sharded world 1) ; part 2: rocprofv3 kernel stats, PMC traffic, SQ counters ; part 3: 512^3
set -u
out=gpurun_out/${1:-r04}
part=${2:-1}
mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = "1" ]; then
python bench.py --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
echo "c2 done" > $out/progress.txt
MG_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/c2_bench_no_pipeline.json 2> /dev/null
MG_NO_MARCH4=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/c2_bench_no_four_stage.json 2> /dev/null
echo "A/B done" >> $out/progress.txt
python bench.py --cells 400 --steps 20 --warmup 5 --no-cpu-baseline --no-divsiggrad > $out/c2_400_bench.json 2> $out/c2_400_bench.err
echo "c2-400 done" >> $out/progress.txt
python bench.py --force-sharded-path --no-cpu-baseline --steps 20 --warmup 5 > $out/c2_sharded_w1_bench.json 2> $out/c2_sharded_w1.err
python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_bench.json 2> $out/c5_bench.err
echo "sharded, c5 done" >> $out/progress.txt
fi
if [ "$part" = "2" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/prof_c2.log 2>&1
f=$(find $out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
t=$(find $out/prof_c2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 profiles/summarize_trace.py $t > $out/c2_kernel_by_grid.md
echo "kernel trace done" >> $out/progress.txt
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 scripts/solve_probe.py 256 20 3 > $out/trace.log 2>&1
t=$(find $out/trace -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 scripts/gap_analysis.py $t > $out/c2_gap_analysis.txt
echo "gap analysis done" >> $out/progress.txt
for cells in 256 400; do
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${cells}_$lc -- python3 scripts/pmc_probe.py $cells > $out/pmc_${cells}_$lc.log 2>&1
  f=$(find $out/pmc_${cells}_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_${cells}_$lc.txt
  echo "pmc $cells $c done" >> $out/progress.txt
done
done
bash scripts/pmc_sq4.sh ${1:-r04}/sq4 1024:0 > $out/sq4.log 2>&1
echo "sq done" >> $out/progress.txt
fi
if [ "$part" = "3" ]; then
python bench.py --cells 512 --steps 10 --warmup 2 --no-cpu-baseline --no-generic-pass > $out/c2_512_bench.json 2> $out/c2_512_bench.err
echo "512 done" >> $out/progress.txt
fi
find $out -name "*.csv" -size +1M -delete
find $out -type d -name "prof_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "trace" -prune -exec rm -rf {} \; 2>/dev/null
ls -la $out
