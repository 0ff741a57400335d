#!/bin/bash
# builds multi-feature-vit_amd/build/libmfvit_attnstamp.so: the library with attention_mfma.hip compiled -DMFVIT_ATTN_STAMP (cycle stamps of
# the persistent attention forward, read by tools/attn_stamps.py through MFVIT_LIB).  Diagnostic only - never shipped, never timed.
set -e
R=$(cd $(dirname $0)/.. && pwd)
B=$R/multi-feature-vit_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -DMFVIT_ATTN_STAMP $MFVIT_STAMP_DEFS -c $R/multi-feature-vit_amd/csrc/attention_mfma.hip -o $B/attention_mfma_stamp.o
objs=$(ls $B/*.o | grep -v "attention_mfma")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmfvit_attnstamp.so $objs $B/attention_mfma_stamp.o
ls -la $B/libmfvit_attnstamp.so
