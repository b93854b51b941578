# round 6, call 18: lane reduction with the one-wave form for levels of more than 4096 entries: A/B on config 4 / shards /
# plain MSMs against the row / top kernels (DG16_NO_LANE_REDUCE=1)
tag=${1:-r6y2}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for rep in 1 2; do
  for v in 0 1; do
    echo "== DG16_NO_LANE_REDUCE=$v" >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 4,8 2>&1 | grep "^world" >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bn254 1 10,12,14,15 2>&1 | tail -4 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bn254 2 10,14 2>&1 | tail -2 >> $O/ab_lane_reduce.txt
    DG16_NO_LANE_REDUCE=$v timeout 120 python tools/msm_small_probe.py bls12_377 1 10,14,15 2>&1 | tail -3 >> $O/ab_lane_reduce.txt
  done
done
cat $O/ab_lane_reduce.txt
