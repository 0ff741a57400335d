#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <tag> <op> [precision] ; writes gpurun_out/pmc_<tag>/pass*/ csv + summary
set -e
tag=$1; op=$2; prec=${3:-bf16}
R=$(pwd)
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python3 $R/tools/one_op.py $op 4 $prec > $out/pass$i.log 2>&1 || { rc=$?; echo "pass $i failed (rc $rc): not starting further passes on this box" >&2; tail -20 $out/pass$i.log >&2; exit $rc; }
done
cd $R
python3 tools/pmc_summary.py $out
