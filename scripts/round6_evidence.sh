#!/bin/bash
# Round-6 evidence on one MI355X box.  Output: gpurun_out/$1/   usage (through gpurun): bash scripts/round6_evidence.sh r06 <part>
#  part 1  the driver's line (C2 + C5 leg + generic-CSR pass + div-sigma-grad leg + C3 leg + CPU baseline) and the bare default run
#  part 2  rocprofv3 kernel stats / kernel-by-grid table of the same command, PMC traffic of the fine-level kernels
#  part 3  sharded path: a world of one (+ the single-GPU path in the same job), dry ranks of 8 / 4 / 2, 512^3 on one GPU, strong ceiling
#  part 4  C5 as its own workload, 400^3, C3 at 128^3
set -u
out=gpurun_out/${1:-r06}
part=${2:-1}
mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = "1" ]; then
python bench.py --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
echo "c2 done" > $out/progress.txt
python bench.py > $out/c2_bench_default_run.json 2> /dev/null
echo "default run done" >> $out/progress.txt
fi
if [ "$part" = "2" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generic-pass --no-divsiggrad --no-c5-leg --no-c3-leg > $out/prof_c2.log 2>&1
f=$(find $out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
t=$(find $out/prof_c2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 profiles/summarize_trace.py $t > $out/c2_kernel_by_grid.md
echo "kernel trace done" >> $out/progress.txt
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_256_$lc -- python3 scripts/pmc_probe.py 256 > $out/pmc_256_$lc.log 2>&1
  f=$(find $out/pmc_256_$lc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 profiles/summarize_pmc.py $f > $out/pmc_256_$lc.txt
  echo "pmc $c done" >> $out/progress.txt
done
fi
if [ "$part" = "3" ]; then
python bench.py --force-sharded-path --steps 20 --warmup 5 > $out/c2_sharded_w1_ghost.json 2> $out/c2_sharded_w1_ghost.err
echo "world-1 done" > $out/progress3.txt
python bench.py --ghost-dry 0/8 --steps 20 --warmup 5 > $out/c4_dry_rank0_of_8.json 2> $out/c4_dry_rank0.err
python bench.py --ghost-dry 7/8 --steps 20 --warmup 5 > $out/c4_dry_rank7_of_8.json 2> $out/c4_dry_rank7.err
python bench.py --ghost-dry 1/2 --steps 20 --warmup 5 > $out/c4_dry_rank1_of_2.json 2> /dev/null
python bench.py --ghost-dry 3/4 --steps 20 --warmup 5 > $out/c4_dry_rank3_of_4.json 2> /dev/null
echo "dry ranks done" >> $out/progress3.txt
python bench.py --cells 512 --steps 20 --warmup 3 --no-cpu-baseline --no-generic-pass --no-divsiggrad --no-c5-leg --no-c3-leg > $out/c2_512_bench.json 2> $out/c2_512_bench.err
echo "512 done" >> $out/progress3.txt
python3 scripts/strong_ceiling.py $out > $out/strong_ceiling.json
fi
if [ "$part" = "4" ]; then
python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $out/c5_bench.json 2> $out/c5_bench.err
echo "c5 done" > $out/progress4.txt
python bench.py --cells 400 --steps 20 --warmup 5 --no-cpu-baseline --no-divsiggrad --no-generic-pass --no-c5-leg --no-c3-leg > $out/c2_400_bench.json 2> $out/c2_400_bench.err
echo "400 done" >> $out/progress4.txt
python bench.py --workload c3 --cells 128 --steps 10 --warmup 2 --no-cpu-baseline > $out/c3_128_bench.json 2> $out/c3_128_bench.err
echo "c3-128 done" >> $out/progress4.txt
fi
find $out -name "*.csv" -size +1M -delete
find $out -type d -name "prof_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
ls -la $out
