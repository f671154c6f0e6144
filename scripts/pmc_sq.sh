#!/bin/bash
# SQ-side PMC passes of any kernel of a probe run.  usage: bash scripts/pmc_sq.sh <outdir under gpurun_out> <kernel name pattern> <probe.py> [args...]
# e.g.  bash scripts/pmc_sq.sh r06/sq27 march27 scripts/pmc_probe.py 256          (level 2 of C2: the 27-point marching form)
#       bash scripts/pmc_sq.sh r06/sqband march3 scripts/band_ab.py 3:0            (div-sigma-grad-like band form of the two-stage pass)
# (counters in their own runs, --kernel-trace only beside --pmc: the pool refuses --pmc with the trace domains)
set -u
out=gpurun_out/$1
pat=$2
shift 2
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 profiles/summarize_pmc.py $f | grep -E "$pat" >> $out/sq.txt
  echo "pass $i done" >> $out/progress.txt
done
find $out -type d -name "p[0-9]*" -prune -exec rm -rf {} \; 2>/dev/null
cat $out/sq.txt
