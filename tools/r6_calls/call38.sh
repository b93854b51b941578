# round 6, call 38: the top-window rule for short tables (msm_window_bits with the scalar width) -- tests that build tables,
# then default against the widths the old rule picked
tag=${1:-r6vv}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_hdist.py tests/test_gpu_dist.py tests/test_gpu_two_rank.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
T=$O/top_window_rule.txt
for c in 0 16; do
  echo "config 4, table c=$c" >> $T
  if [ $c = 0 ]; then unset DG16_MSM_TABLE_C; else export DG16_MSM_TABLE_C=$c; fi
  timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $T
done
unset DG16_MSM_TABLE_C
echo "bn254 2^16 shards, default" >> $T
DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 16 10 bn254 1,2,4,8 2>&1 | grep "^world" >> $T
echo "bn254 2^20 shards, default" >> $T
DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 4,8 2>&1 | grep "^world" >> $T
for c in 0 14 15 16; do
  echo "bls12_381 2^16 shards, table c=$c" >> $T
  if [ $c = 0 ]; then unset DG16_MSM_TABLE_C; else export DG16_MSM_TABLE_C=$c; fi
  DG16_OVERLAP=1 timeout 160 python tools/shard_timing.py 16 5 bls12_381 1,2,4,8 2>&1 | grep "^world" >> $T
done
unset DG16_MSM_TABLE_C
cat $T
