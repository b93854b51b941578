# round 6, call 36: table window width of short keys (DG16_MSM_TABLE_C) again, now that small bucket sets reduce in lane form
tag=${1:-r6tt}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for c in 0 11 12 13 14 15 16; do
  echo "table c=$c" >> $O/table_window_small_keys.txt
  if [ $c = 0 ]; then unset DG16_MSM_TABLE_C; else export DG16_MSM_TABLE_C=$c; fi
  timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/table_window_small_keys.txt
done
for c in 0 13 14 15 16 17; do
  echo "table c=$c" >> $O/table_window_small_keys.txt
  if [ $c = 0 ]; then unset DG16_MSM_TABLE_C; else export DG16_MSM_TABLE_C=$c; fi
  DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/table_window_small_keys.txt
  DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 4 2>&1 | grep "^world" >> $O/table_window_small_keys.txt
done
cat $O/table_window_small_keys.txt
