# A/B of the 14-limb G2 accumulation: round 4's straight-line loop (DG16_G2_14LIMB=inline) against the step loop (default)
O=gpurun_out/r5b; mkdir -p $O
for v in inline steps inline steps; do
  e=""; [ $v = inline ] && e="DG16_G2_14LIMB=inline"
  echo "== $v" >> $O/ab.txt
  env $e CURVE=bls12_381 python tools/perf_probe.py msm2 20 5 2>&1 | tail -1 >> $O/ab.txt
  env $e python bench.py --curve bls12_381 --log-m 20 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('proof ms', d['ms_per_step'], 'single', d['single_proof_ms'], 'G2 acc ms', d['roofline']['kernel_ms'], 'g1 acc', d['g1_accumulate_ms'])" >> $O/ab.txt
done
env CURVE=bls12_377 python tools/perf_probe.py msm2 18 3 2>&1 | tail -1 >> $O/ab.txt
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py -m gpu -q -x -k "bls" 2>&1 | tail -5) > $O/gputest_bls.txt
cat $O/ab.txt; tail -3 $O/gputest_bls.txt
