#!/bin/bash
# usage: bash tools/kstats.sh <tag> <python script + args>   -> prints per-kernel average duration (rocprofv3 --kernel-trace --stats)
tag=$1; shift
R=$(pwd)
script=$1; shift
[[ $script != /* ]] && script=$R/$script      # rocprofv3 runs from /tmp: make the script path absolute
out=$R/gpurun_out/ks_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$script" "$@" > $out/log.txt 2>&1
cd $R
python3 - $out <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"]
        if "mfvit" in n:
            print(f'{float(row["AverageNs"])/1e3:9.1f} us  x{row["Calls"]:>5s}  {n[:110]}')
PY
