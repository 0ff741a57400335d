#!/bin/bash
# usage (GPU box, repo root): bash tools/tile_store_policy.sh <op>: HBM read traffic (FETCH_SIZE) and kernel duration of one tile-GEMM op
# under the four cache policies of its output stores (MFVIT_NT_STORE = 0 plain, 1 nt (default), 2 sc0 sc1 nt, 3 sc1)
set -e
op=${1:-fc1}
R=$(pwd); out=$R/gpurun_out/storepol_$op; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3; do
  MFVIT_NT_STORE=$s rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/s$s -- python3 $R/tools/one_op.py $op 6 bf16x3 > $out/s$s.log 2>&1
  MFVIT_NT_STORE=$s rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$s -- python3 $R/tools/one_op.py $op 12 bf16x3 > $out/t$s.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob
for s in range(4):
    f = [x for x in glob.glob("$out/s%d/**/*counter_collection.csv" % s, recursive=True)]
    v = [float(r["Counter_Value"]) for x in f for r in csv.DictReader(open(x)) if "gemm_nt_tile" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    t = [x for x in glob.glob("$out/t%d/**/*kernel_trace.csv" % s, recursive=True)]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for x in t for r in csv.DictReader(open(x)) if "gemm_nt_tile" in r["Kernel_Name"]]
    d = sorted(d)[: max(1, len(d) - 2)]
    print(f"$op MFVIT_NT_STORE={s}: HBM read {2 * sum(v) / len(v) / 1024:7.1f} MB per launch, duration median {d[len(d) // 2]:6.1f} us (n={len(d)})")
PY
