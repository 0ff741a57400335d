#!/bin/bash
# On the GPU box, from the repo root: bash tools/profile_bench.sh <round-tag>
#   1. rocprofv3 --kernel-trace --stats of the contract bench  -> gpurun_out/prof_<tag>/kernel_stats.csv
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE)        -> gpurun_out/prof_<tag>/hbm_traffic_{per_launch,by_class}.json
set -e
tag=$1
R=$(pwd)
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${2:+--precision $2} > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${2:+--precision $2} > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${2:+--precision $2} > $out/write.log 2>&1
cd $R
python3 tools/profile_bench_summary.py $out ${2:-bf16x3}
# 3. kernel durations of the serialized mode (what roofline.avg_us is checked against)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${2:+--precision $2} --serialize-streams > $out/stats_serial.log 2>&1
cd $R
for f in $(find $out/stats_serial -name "*kernel_stats.csv"); do cp $f $out/kernel_stats_serialized.csv; done
