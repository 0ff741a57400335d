#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_mem.sh <tag> <op> [precision]: memory-side counters of ONE op (L2 hit rate, TCP stall, TA busy) + MFMA busy
set -e
tag=$1; op=$2; prec=${3:-bf16x3}
R=$(pwd)
out=$R/gpurun_out/pmcmem_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for P in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python3 $R/tools/one_op.py $op 4 $prec > $out/pass$i.log 2>&1 || { echo "pass $i failed" >&2; tail -5 $out/pass$i.log >&2; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $out
