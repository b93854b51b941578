# round 6, call 37: table window width against key size: shards of a 2^16 proof (2^16 .. 2^13 points per rank)
tag=${1:-r6uu}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for c in 0 11 12 13 14 15 16; do
  echo "table c=$c" >> $O/table_window_by_size.txt
  if [ $c = 0 ]; then unset DG16_MSM_TABLE_C; else export DG16_MSM_TABLE_C=$c; fi
  DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 16 10 bn254 1,2,4,8 2>&1 | grep "^world" >> $O/table_window_by_size.txt
done
cat $O/table_window_by_size.txt
