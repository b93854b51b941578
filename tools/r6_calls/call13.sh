# round 6, call 13: lane chains for the fourteen-limb G1 groups: GPU suite of the touched paths, A/B against
# -DDG16_NO_LANE_CHAINS on the shapes of msm_bench.rs (BLS12-377 G1) and on a BLS12-381 proof
tag=${1:-r6r}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py tests/test_gpu_hdist.py tests/test_gpu_two_rank.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in lane nolane; do
    l="DG16_X=0"; [ $v = nolane ] && l="DG16_LIB=$D/libdg16_nolane.so"
    echo "== $v" >> $O/ab_lane_chains.txt
    env $l timeout 120 python tools/msm_small_probe.py bls12_377 1 10,12,14,16,19 2>&1 | tail -5 >> $O/ab_lane_chains.txt
    env $l timeout 120 python tools/msm_small_probe.py bls12_381 1 12,20 2>&1 | tail -2 >> $O/ab_lane_chains.txt
    env $l timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_lane_chains.txt
  done
done
cat $O/ab_lane_chains.txt
