# round-3 A/B sweep of the resident-key prover on one box (same-call A/B: boxes differ by +-5 %)
for cfg in "" "DG16_SORT_OVERLAP=0" "" "DG16_SORT_OVERLAP=0"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/perf_probe.py prove 20 10 2>/dev/null | grep groth16
done
for lg in 16 18 20 22; do python tools/perf_probe.py hpoly $lg 10 2>/dev/null | grep h_poly; python tools/perf_probe.py ntt $lg 10 2>/dev/null | grep ntt; done
