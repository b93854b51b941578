# round 6, call 41: the 16-entry segments of plain 2^20..2^21-entry MSMs as the default; lane finalize against the lone-lane
# one around its threshold; MSM / d_msm GPU tests
tag=${1:-r6yy}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/plain_msm_seg16_default.txt
for nl in 0 1; do
  if [ $nl = 0 ]; then unset DG16_NO_LANE_FINALIZE; else export DG16_NO_LANE_FINALIZE=1; fi
  echo "DG16_NO_LANE_FINALIZE=$nl" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_377 1 13,14,15,16,17,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 1 14,16,17 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 2 14,16 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bn254 2 14,16 10 2>&1 | grep "2^" >> $T
done
unset DG16_NO_LANE_FINALIZE
cat $T
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cat $O/tests.txt
