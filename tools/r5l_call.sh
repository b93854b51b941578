O=gpurun_out/r5l; mkdir -p $O
for v in 3 4 3 4; do
  echo "== 2^20 DG16_MSM_SEG_LOG_EXP=$v" >> $O/ab.txt
  DG16_MSM_SEG_LOG_EXP=$v python bench.py --curve bls12_381 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
for v in 4 5 4 5; do
  echo "== 2^22 DG16_MSM_SEG_LOG_EXP=$v" >> $O/ab.txt
  DG16_MSM_SEG_LOG_EXP=$v python bench.py --curve bls12_381 --log-m 22 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls 2^22 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
for v in 4 5 4 5; do
  echo "== bn254 2^22 DG16_MSM_SEG_LOG_EXP=$v" >> $O/ab.txt
  DG16_MSM_SEG_LOG_EXP=$v python bench.py --log-m 22 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn254 2^22 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
cat $O/ab.txt
