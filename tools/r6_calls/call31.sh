# round 6, call 31: B's G2 finalize on its reduction stream (under A's accumulation) instead of the main stream, same library
tag=${1:-r6oo}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(DG16_G2_FINALIZE_SIDE=1 timeout 600 python -X faulthandler -m pytest tests/test_gpu_prover.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -4) > $O/pytest_side.txt
cat $O/pytest_side.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== DG16_G2_FINALIZE_SIDE=$v" >> $O/ab_g2_finalize_side.txt
    DG16_G2_FINALIZE_SIDE=$v DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 1,8 2>&1 | grep "^world" >> $O/ab_g2_finalize_side.txt
    DG16_G2_FINALIZE_SIDE=$v DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 1 2>&1 | grep "^world" >> $O/ab_g2_finalize_side.txt
    DG16_G2_FINALIZE_SIDE=$v timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_g2_finalize_side.txt
  done
done
cat $O/ab_g2_finalize_side.txt
