# round 6, call 42: segment length (DG16_MSM_SEG_LOG) of the short TABLE launches inside proofs: config 4, 8- and 4-shard ranks
tag=${1:-r6zy}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/seg_log_short_proofs.txt
for sl in 0 3 4 5 0; do
  if [ $sl = 0 ]; then unset DG16_MSM_SEG_LOG; else export DG16_MSM_SEG_LOG=$sl; fi
  echo "seg_log=$sl" >> $T
  timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $T
  DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 2,4,8 2>&1 | grep "^world" >> $T
  DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 16 10 bn254 1 2>&1 | grep "^world" >> $T
  DG16_OVERLAP=1 timeout 160 python tools/shard_timing.py 20 5 bls12_381 8 2>&1 | grep "^world" >> $T
done
cat $T
