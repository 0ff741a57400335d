set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh r05_attn_fwd attn_fwd bf16x3 > gpurun_out/r05_pmc_attn_fwd.txt 2>&1
bash tools/pmc.sh r05_attn_bwd attn_bwd bf16x3 > gpurun_out/r05_pmc_attn_bwd.txt 2>&1
for op in qkv fc1 fc2_dgrad; do bash tools/pmc_mem.sh r05_$op $op bf16x3 > gpurun_out/r05_pmcmem_$op.txt 2>&1; done
for op in fc1 fc2_dgrad; do MFVIT_NT_WIDE=1 bash tools/pmc_mem.sh r05_${op}_wide $op bf16x3 > gpurun_out/r05_pmcmem_${op}_wide.txt 2>&1; done
echo PMC_DONE
