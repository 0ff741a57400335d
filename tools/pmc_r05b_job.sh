# the second PMC batch of round 5 (final kernels): instruction mix / MFMA busy / HBM bytes of the GEMM classes, isolated launches at the bench shape
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for op in qkv fc1 row_fc2 rowb_fc1 wgrad_fc1; do bash tools/pmc.sh r05b_$op $op bf16x3 > gpurun_out/r05b_pmc_$op.txt 2>&1; done
echo PMC_DONE
