#!/bin/bash
# SQ / TCC counters of the transfer kernels INSIDE the solve loop (their source vectors as cold as the cycle leaves them).
# usage (through gpurun): bash scripts/pmc_sq_rp.sh <outdir>
set -u
out=gpurun_out/${1:-sqrp}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_FLAT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 scripts/prolong_ab.py - > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 profiles/summarize_pmc.py $f | grep -E "winp|lane_spmv<0|march3|tile_spmv" >> $out/sq.txt
  echo "pass $i done" >> $out/progress.txt
done
find $out -type d -name "p[0-9]*" -prune -exec rm -rf {} \; 2>/dev/null
cat $out/sq.txt
