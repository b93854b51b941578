# Round 4: DG16_F_OVERLAP_TAIL -- the queue test, the prover tests, and the bench with / without the overlap (same box).
O=gpurun_out/r4u; mkdir -p $O
(timeout 500 python -X faulthandler -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "queue or medium or bigint" -o faulthandler_timeout=200 2>&1 | tail -15) > $O/gputest.txt
tail -3 $O/gputest.txt
for pass in 1 2; do
  for mode in "" "--no-overlap"; do
    timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline $mode 2> $O/bench.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('overlap' if '$mode' == '' else 'no-overlap', 'ms_per_step', round(d['ms_per_step'], 3), 'single_proof_ms', round(d['single_proof_ms'], 3), d.get('parity_check'))" >> $O/overlap_ab.txt
  done
done
timeout 200 python bench.py --curve bls12_381 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>> $O/bench.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('bls12_381 overlap ms_per_step', round(d['ms_per_step'], 3), 'single_proof_ms', round(d['single_proof_ms'], 3))" >> $O/overlap_ab.txt
cat $O/overlap_ab.txt; tail -5 $O/bench.err
