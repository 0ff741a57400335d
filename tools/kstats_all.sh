#!/bin/bash
# usage: bash tools/kstats_all.sh <tag> <python script + args>   -> per-kernel totals of every kernel (rocprofv3 --kernel-trace --stats)
tag=$1; shift
R=$(pwd)
out=$R/gpurun_out/ks_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/log.txt 2>&1
cd $R
python3 - $out <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{float(r["TotalDurationNs"])/1e6:8.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%  avg {float(r["AverageNs"])/1e3:8.1f} us x{r["Calls"]:>5s}  {r["Name"][:100]}')
PY
