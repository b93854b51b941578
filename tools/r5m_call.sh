O=gpurun_out/r5m; mkdir -p $O
(timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -o faulthandler_timeout=400 2>&1 | grep -E "passed|failed|Error|error" | tail -5) > $O/gputest.txt
cat $O/gputest.txt
for v in 17 16 18 17 16 18; do
  echo "== DG16_MSM_TABLE_C=$v" >> $O/ab.txt
  DG16_MSM_TABLE_C=$v python bench.py --curve bls12_381 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3), d['config']['key_window_bits'])" >> $O/ab.txt
done
python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn254 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
cat $O/ab.txt
