#!/bin/bash
# SQ-side PMC passes of the four-stage pass (one variant).  usage: bash scripts/pmc_sq4.sh <outdir> <variant nt:tiles_x[:k1]>
set -u
out=gpurun_out/${1:-sq4}
var=${2:-1024:0}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 scripts/march4_ab.py 256 $var > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 profiles/summarize_pmc.py $f | grep -E "march4" >> $out/sq.txt
done
find $out -type d -name "p[0-9]*" -prune -exec rm -rf {} \; 2>/dev/null
cat $out/sq.txt
