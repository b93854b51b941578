O=gpurun_out/r5i; mkdir -p $O
for v in 0 3 2 0 3 1; do
  echo "== DG16_ROW_CHUNK_EXP=$v" >> $O/ab.txt
  DG16_ROW_CHUNK_EXP=$v python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn254 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
for v in 0 3 0 3; do
  echo "== DG16_ROW_CHUNK_EXP=$v" >> $O/ab.txt
  DG16_ROW_CHUNK_EXP=$v python bench.py --curve bls12_381 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
cat $O/ab.txt
