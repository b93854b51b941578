# geometry sweep of the resident-key prover on one box (same-call A/B: boxes differ by +-5 %)
for cfg in "" "DG16_MSM_TABLE_C=19" "DG16_MSM_TABLE_C=20" "DG16_MSM_TABLE_C=20 DG16_MSM_SEG_LOG=3" "DG16_MSM_TABLE_C=16" "DG16_MSM_SEG_LOG=5" ""; do
  echo "== $cfg"; env $cfg timeout 120 python tools/shard_timing.py 20 10 bn254 1 2>/dev/null | grep world
done
