# round 6, call 29: the 14-limb G1 reductions (row / top / lone-lane finalize) with out-of-line products: parity, A/B against
# the previous library (inline), with and without the lane-form finalize
tag=${1:-r6mm}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_hdist.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for v in outline inline; do
    l="DG16_X=0"; [ $v = inline ] && l="DG16_LIB=$D/libdg16_prev.so"
    for nf in 0 1; do
      echo "== $v DG16_NO_LANE_FINALIZE=$nf" >> $O/ab_outline_bls_g1.txt
      env $l DG16_NO_LANE_FINALIZE=$nf timeout 200 python tools/msm_small_probe.py bls12_377 1 13,16,17,18,19 2>&1 | tail -5 >> $O/ab_outline_bls_g1.txt
    done
    echo "== $v" >> $O/ab_outline_bls_g1.txt
    env $l timeout 200 python tools/msm_small_probe.py bls12_381 1 20 2>&1 | tail -1 >> $O/ab_outline_bls_g1.txt
    env $l timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_outline_bls_g1.txt
    env $l DG16_OVERLAP=1 timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_outline_bls_g1.txt
  done
done
cat $O/ab_outline_bls_g1.txt
