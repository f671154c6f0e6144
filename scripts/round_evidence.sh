#!/bin/bash
# Round-end evidence on one MI355X box: bench lines, rocprofv3 kernel stats, PMC passes.  Output: gpurun_out/$1/
# usage (through gpurun): bash scripts/round_evidence.sh r02e
set -u
out=gpurun_out/${1:-evidence}
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
python bench.py --workload c5 --steps 10 --warmup 2 > $out/c5_bench.json 2> $out/c5_bench.err
python bench.py --force-sharded-path --no-cpu-baseline --steps 20 --warmup 5 > $out/c2_sharded_w1_bench.json 2> $out/c2_sharded_w1.err
python scripts/lu_coarse_time.py > $out/lu_coarse.txt 2>&1
(cd profiles/calib && ./triad_calib) > $out/triad_calib.txt 2>&1
MG_NO_MARCH2=1 MG_NO_TILE_LANE=1 MG_NO_WINP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/c2_bench_single_stage.json 2> $out/c2_bench_single_stage.err
python bench.py --workload c3 --cells 128 --steps 10 --warmup 2 --no-cpu-baseline > $out/c3_128_bench.json 2> $out/c3_128_bench.err
python scripts/diag_host_api.py 2>&1 | grep "host API" > $out/host_api.txt
python scripts/diag_pcg.py 2>&1 | grep "ms per" > $out/krylov.txt
# rocprofv3: the program itself after "--"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass > $out/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c5 -- python3 bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_c5.log 2>&1
for w in c2 c5; do
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${w}_kernel_stats.csv
  t=$(find $out/prof_$w -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 profiles/summarize_trace.py $t > $out/${w}_kernel_by_grid.md
done
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_c2_$lc -- python3 scripts/pmc_probe.py 256 > $out/pmc_c2_$lc.log 2>&1
  f=$(find $out/pmc_c2_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_c2_$lc.txt
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_c5_$lc -- python3 scripts/pmc_probe.py 256 16 > $out/pmc_c5_$lc.log 2>&1
  f=$(find $out/pmc_c5_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_c5_$lc.txt
done
find $out -name "*.csv" -size +1M -delete
find $out -type d -name "prof_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
ls -la $out
