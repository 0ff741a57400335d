set -e
echo "baseline 2 WG/CU BKB128 deep, no epilogue:"
MFVIT_ABLATE_BITS=8 timeout -k 10 200 bash tools/kstats.sh o0 tools/one_op.py qkv 20 bf16x3 | grep tile
for occ in 22 30 32 40 42; do
  echo "occ code $occ:"
  MFVIT_TILE_OCC=$occ timeout -k 10 200 bash tools/kstats.sh o$occ tools/one_op.py qkv 20 bf16x3 | grep tile
done
rm -rf gpurun_out/ks_*
