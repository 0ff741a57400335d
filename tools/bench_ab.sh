#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_ab.sh <variant .so> [rounds] [bench args]: the contract bench alternating between the shipping library and a
# variant library (MFVIT_LIB) on ONE box: ms per step and the per-class times of the serialized pass
V=$1; N=${2:-2}; shift; shift
for i in $(seq $N); do
  for which in ship variant; do
    if [ $which = variant ]; then export MFVIT_LIB=$V; else unset MFVIT_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
pc=d['roofline']['serialized_pass']['per_class']
print('$which', 'ms_per_step', round(d['ms_per_step'],3), 'serialized', d['roofline']['serialized_pass']['ms_per_step'], ' '.join(f\"{k.replace('gemm_','').replace('attention','attn')}={v['avg_us']}\" for k,v in pc.items() if 'xattn' not in k and k!='other'))"
  done
done
