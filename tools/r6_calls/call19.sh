# round 6, call 19: lane-form levels in place of msm_top_kernel (msm_lane_top): MSM parity, A/B (DG16_NO_LANE_TOP=1)
tag=${1:-r6aa}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in 0 1; do
    echo "== DG16_NO_LANE_TOP=$v" >> $O/ab_lane_top.txt
    DG16_NO_LANE_TOP=$v timeout 120 python tools/msm_small_probe.py bn254 1 16,18,20 2>&1 | tail -3 >> $O/ab_lane_top.txt
    DG16_NO_LANE_TOP=$v timeout 120 python tools/msm_small_probe.py bn254 2 16,18,20 2>&1 | tail -3 >> $O/ab_lane_top.txt
    DG16_NO_LANE_TOP=$v timeout 120 python tools/msm_small_probe.py bls12_377 1 16,17,18,19 2>&1 | tail -4 >> $O/ab_lane_top.txt
    DG16_NO_LANE_TOP=$v timeout 120 python tools/msm_small_probe.py bls12_381 1 20 2>&1 | tail -1 >> $O/ab_lane_top.txt
    DG16_NO_LANE_TOP=$v DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 1,2 2>&1 | grep "^world" >> $O/ab_lane_top.txt
  done
done
cat $O/ab_lane_top.txt
