# round-3 A/B sweep of the resident-key prover on one box (same-call A/B: boxes differ by +-5 %)
for cfg in "" "DG16_FINALIZE=2" "DG16_FINALIZE_MAIN=1" "DG16_FINALIZE_MAIN=1 DG16_FINALIZE=2" "DG16_FINALIZE_MAIN=1 DG16_FINALIZE=2 DG16_MSM_SEG_LOG=5" "DG16_FINALIZE=2 DG16_MSM_SEG_LOG=5" "DG16_FINALIZE_MAIN=1 DG16_MSM_SEG_LOG=5" ""; do
  echo "== $cfg"; env $cfg timeout 120 python tools/perf_probe.py prove 20 10 2>/dev/null | grep groth16
done
