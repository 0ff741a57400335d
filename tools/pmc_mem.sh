#!/bin/bash
# memory-side counters of one op: bash tools/pmc_mem.sh <tag> <op> [precision]  (L1->L2 read latency, L2 hit rate, TA / TCC busy, stalls)
set -e
tag=$1; op=$2; prec=${3:-bf16}
R=$(pwd)
out=$R/gpurun_out/pmcm_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum"
P2="TA_BUSY_avr TCC_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P3="TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
P4="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
P5="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_WRITE_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python3 $R/tools/one_op.py $op 4 $prec > $out/pass$i.log 2>&1 || { rc=$?; echo "pass $i failed (rc $rc)" >&2; tail -20 $out/pass$i.log >&2; exit $rc; }
done
cd $R
python3 tools/pmc_summary.py $out
rm -rf $out/pass*/
