#!/bin/bash
# A/B of the stream / pairing defaults per precision and shape: prints pairs/s (or samples/s) per variant
cd $(dirname $0)/..
run() { # tag, env..., -- bench args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  v=$(env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
  echo "$tag: $v"
}
for cfg in "ca224_fp16 --precision fp16" "ca384_fp16 --precision fp16 --img 384 --batch 32" "moco_fp16 --workload moco --precision fp16" "ca224_bf16 --precision bf16" "single_x3 --workload single"; do
  set -- $cfg; name=$1; shift
  run "$name default" X=1 -- "$@"
  run "$name wgrad_stream=1" MFVIT_WGRAD_STREAM=1 -- "$@"
  run "$name wgrad_stream=0 pair=0" MFVIT_TN_PAIR=0 -- "$@"
done
