# round 6, call 7: G1 tree experiments (fixed print); table window sweep for small keys (config 4, 8-way shard)
tag=${1:-r6g}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
for rep in 1 2; do
  for v in product g1outline g1into g1b128; do
    l="DG16_X=0"; [ $v != product ] && l="DG16_LIB=$D/libdg16_$v.so"
    env $l timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'A acc ms', round(d['g1_accumulate_ms'],4), 'G2 acc', round(d['roofline']['kernel_ms'],3))" >> $O/g1_tree_experiments.txt
  done
done
cat $O/g1_tree_experiments.txt
for c in 0 12 13 14 15 16; do
  e="DG16_X=0"; [ $c != 0 ] && e="DG16_MSM_TABLE_C=$c"
  echo "table c=$c" >> $O/table_c_small_keys.txt
  env $e timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/table_c_small_keys.txt
done
for c in 0 14 15 16 17; do
  e="DG16_X=0"; [ $c != 0 ] && e="DG16_MSM_TABLE_C=$c"
  echo "table c=$c" >> $O/table_c_small_keys.txt
  env $e DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/table_c_small_keys.txt
done
cat $O/table_c_small_keys.txt
