# round 6, call 30: the G2 groups' throughput finalize (LPB lanes per bucket + tree, accumulator in LDS columns) for the
# 14-limb G1 groups in place of the lone-lane finalize (DG16_FINALIZE14_LPB=0 / 2 / 4): parity under 4, then A/B
tag=${1:-r6nn}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(DG16_FINALIZE14_LPB=4 DG16_NO_LANE_FINALIZE=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_msm.py -q -m gpu -x -k "bls or 377 or 381 or matches_oracle or giant" 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -6) > $O/pytest_lpb4.txt
cat $O/pytest_lpb4.txt
for rep in 1 2; do
  for lpb in 0 2 4; do
    echo "== DG16_FINALIZE14_LPB=$lpb" >> $O/ab_finalize14.txt
    DG16_FINALIZE14_LPB=$lpb DG16_NO_LANE_FINALIZE=1 timeout 200 python tools/msm_small_probe.py bls12_377 1 13,16 2>&1 | tail -2 >> $O/ab_finalize14.txt
    DG16_FINALIZE14_LPB=$lpb timeout 200 python tools/msm_small_probe.py bls12_377 1 17,18,19 2>&1 | tail -3 >> $O/ab_finalize14.txt
    DG16_FINALIZE14_LPB=$lpb timeout 200 python tools/msm_small_probe.py bls12_381 1 20 2>&1 | tail -1 >> $O/ab_finalize14.txt
    DG16_FINALIZE14_LPB=$lpb timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_finalize14.txt
    DG16_FINALIZE14_LPB=$lpb DG16_OVERLAP=1 timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_finalize14.txt
  done
done
cat $O/ab_finalize14.txt
