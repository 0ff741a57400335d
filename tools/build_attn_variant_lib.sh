#!/bin/bash
# builds multi-feature-vit_amd/build/libmfvit_attnvar_$1.so with attention_mfma.hip compiled with extra defines $2.. (experiments only)
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/multi-feature-vit_amd/build
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -c $R/multi-feature-vit_amd/csrc/attention_mfma.hip -o $B/attention_mfma_var_$tag.o
objs=$(ls $B/*.o | grep -v "attention_mfma.o" | grep -v attention_mfma_stamp | grep -v attention_mfma_var)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmfvit_attnvar_$tag.so $objs $B/attention_mfma_var_$tag.o
