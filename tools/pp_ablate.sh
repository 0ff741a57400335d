#!/bin/bash
# compile-time ablation of the ping-pong GEMM: builds one library per bit set (on the build host), tools/pp_ablate_run.sh times them on the GPU
R=$(cd $(dirname $0)/.. && pwd)
for b in "$@"; do
  ( MFVIT_TRACE_DEFS="-DMFVIT_PP_NOTICKS -DMFVIT_PP_ABL=$b" bash $R/tools/build_trace_lib.sh _abl$b > /dev/null 2>&1 || echo "build $b failed" ) &
done
wait
ls $R/multi-feature-vit_amd/build/ | grep abl
