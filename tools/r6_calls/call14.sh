# round 6, call 14: the lane-form bucket reduction of small bucket sets (msm_lane_reduce_kernel): GPU suite of the touched
# paths, then A/B against the row / top kernels in the same library (DG16_NO_LANE_REDUCE=1)
tag=${1:-r6s}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py tests/test_gpu_hdist.py tests/test_gpu_two_rank.py tests/test_gpu_lane29.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in 0 1; do
    echo "== DG16_NO_LANE_REDUCE=$v" >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 4,8 2>&1 | grep "^world" >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bn254 1 10,12,14 2>&1 | tail -3 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bn254 2 10,12,14 2>&1 | tail -3 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bls12_377 1 10,12,14 2>&1 | tail -3 >> $O/ab_lane_reduce.txt
  done
done
cat $O/ab_lane_reduce.txt
