#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_16bit_configs.sh <tag>: the bench lines WITH the per-class tables of the BASELINE configs that name a plain
# 16-bit dtype (VERDICT r4 task 6): configs[1] single stream bf16 B = 64, configs[3] MoCo slice fp16, configs[4] shape 384^2 fp16 B = 32; plus the bf16x3 lines
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err; tail -c 400 $out/$name.err | grep -v "^\[bench\] gpu" | tail -2; }
run single_bf16 --workload single --precision bf16
run single_x3 --workload single --precision bf16x3
run moco_fp16 --workload moco --precision fp16
run moco_x3 --workload moco --precision bf16x3
run ca384_fp16 --img 384 --batch 32 --precision fp16
run ca224_fp16 --precision fp16
run ca224_bf16 --precision bf16
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/*.json")):
    ls = [l for l in open(f) if l.startswith("{")]
    if not ls:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ls[-1])
    sp = d.get("serialized_pass") or d.get("roofline", {}).get("serialized_pass")
    print(f"{os.path.basename(f):18s} {d['value']:9.1f} {d['unit']:11s} {d['ms_per_step']:7.2f} ms/step  serialized {sp['ms_per_step']:7.2f} ms")
    for k, v in sp["per_class"].items():
        frac = v.get("frac_of_mfma_peak", v.get("frac_of_hbm_peak"))
        print(f"     {k:22s} x{v['launches_per_step']:5.1f} {v['avg_us']:8.1f} us  {v['ms_per_step']:6.3f} ms  frac {frac}")
PY
