#!/bin/bash
# usage: build_variant_lib.sh <name> [defs ...]  ->  multi-feature-vit_amd/build/variants/<name>/libmfvit_hip.so: EVERY source recompiled under the
# given -D switches with the shipping flags (timing / traffic A/B runs on one box, loaded through MFVIT_LIB).  Diagnostic only - never shipped.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/multi-feature-vit_amd/csrc; V=$R/multi-feature-vit_amd/build/variants/$1; shift
mkdir -p $V
for f in $C/*.hip; do
  b=$(basename $f .hip); x=""
  case $b in attention_mfma|attention_tiled|gemm_rowp|gemm_pp) x="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $x "$@" -c $f -o $V/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libmfvit_hip.so $V/*.o
ls -la $V/libmfvit_hip.so
