#!/bin/bash
# usage: build_attn_variant_lib.sh <name> [defs ...]  ->  multi-feature-vit_amd/build/variants/libmfvit_<name>.so: the shipping objects with
# attention_mfma.hip recompiled under the given -D switches (timing experiments, loaded through MFVIT_LIB).  Diagnostic only - never shipped.
set -e
R=$(cd $(dirname $0)/.. && pwd); B=$R/multi-feature-vit_amd/build; V=$B/variants; n=$1; shift
mkdir -p $V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -c $R/multi-feature-vit_amd/csrc/attention_mfma.hip -o $V/attention_mfma_$n.o
objs=$(ls $B/*.o | grep -v "attention_mfma")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libmfvit_$n.so $objs $V/attention_mfma_$n.o
