# round 6, call 33: B's G2 finalize on the side stream for BLS12-381 (14-limb G2: one wave per SIMD) -- A/B in one library
tag=${1:-r6qq}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== DG16_G2_FINALIZE_SIDE=$v" >> $O/ab_g2_finalize_side_bls.txt
    DG16_G2_FINALIZE_SIDE=$v DG16_OVERLAP=1 timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_g2_finalize_side_bls.txt
    DG16_G2_FINALIZE_SIDE=$v DG16_OVERLAP=0 timeout 200 python tools/shard_timing.py 20 5 bls12_381 1 2>&1 | grep "^world" >> $O/ab_g2_finalize_side_bls.txt
  done
done
cat $O/ab_g2_finalize_side_bls.txt
