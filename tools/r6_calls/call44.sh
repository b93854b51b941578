# round 6, call 44: segment length at config 5's size on one GPU (BLS12-381 2^24: 2^27.7 entries per MSM, default 32)
tag=${1:-r6zx}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/seg_log_bls12_381_2e24.txt
for sl in 0 6 7 4; do
  if [ $sl = 0 ]; then unset DG16_MSM_SEG_LOG; else export DG16_MSM_SEG_LOG=$sl; fi
  echo "seg_log=$sl" >> $T
  timeout 600 python bench.py --curve bls12_381 --log-m 24 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],2), 'single', round(d['single_proof_ms'],2), 'g2', round(d['roofline']['kernel_ms'],2), 'g1', round(d['g1_accumulate_ms'],2), d.get('parity_check'))" >> $T
done
cat $T
