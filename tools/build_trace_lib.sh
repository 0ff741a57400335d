#!/bin/bash
# builds multi-feature-vit_amd/build/libmfvit_trace$1.so: the library with gemm_pp.hip compiled -DMFVIT_PP_TRACE $MFVIT_TRACE_DEFS
# (tools/pp_trace.py: cycle stamps; -DMFVIT_PP_NOTICKS -DMFVIT_PP_ABL=<bits>: compile-time ablation, tools/pp_ablate.sh)
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/multi-feature-vit_amd/build
T=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -DMFVIT_PP_TRACE $MFVIT_TRACE_DEFS -c $R/multi-feature-vit_amd/csrc/gemm_pp.hip -o $B/gemm_pp_trace$T.o
objs=$(ls $B/*.o | grep -v gemm_pp.o | grep -v gemm_pp_trace)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmfvit_trace$T.so $objs $B/gemm_pp_trace$T.o
ls -la $B/libmfvit_trace$T.so
