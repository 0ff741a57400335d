#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_env_ab.sh "<VAR=a>" "<VAR=b>" [rounds] [bench args]: the contract bench alternating between two settings of a
# run-time switch of the shipping library on ONE box: ms per step and the per-class times of the serialized pass
A=$1; B=$2; N=${3:-2}; shift; shift; shift
for i in $(seq $N); do
  for which in "$A" "$B"; do
    env $which python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
pc=d['roofline']['serialized_pass']['per_class']
print('$which', 'ms_per_step', round(d['ms_per_step'],3), 'serialized', d['roofline']['serialized_pass']['ms_per_step'], ' '.join(f\"{k.replace('gemm_','').replace('attention','attn')}={v['avg_us']}\" for k,v in pc.items() if 'xattn' not in k and k!='other'), 'parity', d.get('parity_vs_cpu_oracle',{}).get('logits_max_rel_err'))"
  done
done
