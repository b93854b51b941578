# round 6, call 23: lane chains for the 14-limb Fq2 (G2 of BLS12-381 / BLS12-377): the whole GPU suite, then A/B against
# -DDG16_NO_LANE_CHAINS on plain G2 MSMs of both curves and a BLS12-381 proof
tag=${1:-r6gg}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -x -o faulthandler_timeout=400 > $O/gputest_full.txt 2>&1
grep -E "passed|failed|error" $O/gputest_full.txt | tail -3
for rep in 1 2; do
  for v in lane nolane; do
    l="DG16_X=0"; [ $v = nolane ] && l="DG16_LIB=$D/libdg16_nolane.so"
    echo "== $v" >> $O/ab_lane_chains_bls_g2.txt
    env $l timeout 200 python tools/msm_small_probe.py bls12_377 2 10,14,18 2>&1 | tail -3 >> $O/ab_lane_chains_bls_g2.txt
    env $l timeout 200 python tools/msm_small_probe.py bls12_381 2 10,14,20 2>&1 | tail -3 >> $O/ab_lane_chains_bls_g2.txt
    env $l timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_lane_chains_bls_g2.txt
  done
done
cat $O/ab_lane_chains_bls_g2.txt
