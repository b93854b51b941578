O=gpurun_out/r5k; mkdir -p $O
for v in 4 5 6 4 5 6; do
  echo "== DG16_MSM_SEG_LOG_EXP=$v" >> $O/ab.txt
  DG16_MSM_SEG_LOG_EXP=$v python bench.py --curve bls12_381 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
done
cat $O/ab.txt
(timeout 600 python -m pytest tests/test_gpu_prover.py tests/test_gpu_msm.py -m gpu -q -x -k "bls" 2>&1 | grep -E "passed|failed|Error" | tail -3) > $O/gputest_bls.txt
cat $O/gputest_bls.txt
