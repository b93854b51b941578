# round 6, call 28: lane-form finalize of the non-tree groups (BLS12 G1) for small launches: parity, A/B (DG16_NO_LANE_FINALIZE=1)
tag=${1:-r6ll}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in 0 1; do
    echo "== DG16_NO_LANE_FINALIZE=$v" >> $O/ab_lane_finalize.txt
    DG16_NO_LANE_FINALIZE=$v timeout 200 python tools/msm_small_probe.py bls12_377 1 10,11,12,13,14,15,16,17 2>&1 | tail -8 >> $O/ab_lane_finalize.txt
    DG16_NO_LANE_FINALIZE=$v timeout 200 python tools/msm_small_probe.py bls12_381 1 12,14,16 2>&1 | tail -3 >> $O/ab_lane_finalize.txt
  done
done
cat $O/ab_lane_finalize.txt
timeout 300 python tools/dmsm_probe.py 30 > $O/dmsm_sweep.txt 2>&1; tail -10 $O/dmsm_sweep.txt
