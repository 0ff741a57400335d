set -e
bash tools/profile_bench.sh r02 bf16x3 > gpurun_out/prof_r02.log 2>&1
rm -rf gpurun_out/prof_r02/stats gpurun_out/prof_r02/fetch gpurun_out/prof_r02/write gpurun_out/prof_r02/stats_serial
ls gpurun_out/prof_r02
for op in qkv fc1 row_fc2 rowb_fc1 wgrad_fc1 attn_fwd attn_bwd; do
  timeout -k 10 300 bash tools/pmc.sh x3_$op $op bf16x3 > gpurun_out/pmc_x3_$op.txt 2>&1 || { echo "pmc $op failed"; tail -5 gpurun_out/pmc_x3_$op.txt; exit 1; }
  rm -rf gpurun_out/pmc_x3_$op
  echo "pmc $op done"
done
tools/mfma_ceiling 1 random 8 > gpurun_out/ceiling.jsonl; tools/mfma_ceiling 2 random 8 >> gpurun_out/ceiling.jsonl; tools/mfma_ceiling 1 zero 8 >> gpurun_out/ceiling.jsonl
tools/mfma_ceiling 1 random 4 >> gpurun_out/ceiling.jsonl; tools/mfma_ceiling 1 random 2 >> gpurun_out/ceiling.jsonl; tools/mfma_ceiling 1 random 1 >> gpurun_out/ceiling.jsonl
cat gpurun_out/ceiling.jsonl
