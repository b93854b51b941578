# round 6, call 12: lane chains for BN254 G2 as well (Fq2 on rows): GPU suite of the touched paths, A/B against -DDG16_NO_LANE_CHAINS
tag=${1:-r6p}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
(timeout 900 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py tests/test_gpu_hdist.py tests/test_gpu_two_rank.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in lane nolane; do
    l="DG16_X=0"; [ $v = nolane ] && l="DG16_LIB=$D/libdg16_nolane.so"
    echo "== $v" >> $O/ab_lane_chains.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_lane_chains.txt
    env $l DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 1,8 2>&1 | grep "^world" >> $O/ab_lane_chains.txt
    env $l timeout 120 python tools/msm_small_probe.py bn254 2 10,12,14,16,20 2>&1 | tail -5 >> $O/ab_lane_chains.txt
  done
done
cat $O/ab_lane_chains.txt
