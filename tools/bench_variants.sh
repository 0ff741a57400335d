#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_variants.sh <rounds> <variant name> ...: the contract bench with the shipping library and with each variant library of
# build/variants/<name>/ in turn, <rounds> times on ONE box: ms per step, class averages of the serialized pass, the forward-MHSA parts (qkv / core / proj + LN)
N=$1; shift
for i in $(seq $N); do
  for v in ship "$@"; do
    if [ $v = ship ]; then unset MFVIT_LIB; else export MFVIT_LIB=$PWD/multi-feature-vit_amd/build/variants/$v/libmfvit_hip.so; fi
    python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
r=d['roofline']; pc=r['serialized_pass']['per_class']
print('%-8s' % '$v', 'ms_per_step', round(d['ms_per_step'],3), ' '.join(f\"{k.replace('gemm_','').replace('attention','attn')}={v['avg_us']}\" for k,v in pc.items() if 'xattn' not in k and k!='other'), 'mhsa parts', r['fused_mhsa']['parts_avg_us'])"
  done
done
