set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee gpurun_out/gpu_suite.log && python bench.py --steps 20 --warmup 5 > gpurun_out/c2_bench_final.json 2> gpurun_out/c2_bench_final.err; tail -c 600 gpurun_out/c2_bench_final.json
