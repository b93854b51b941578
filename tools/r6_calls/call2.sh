# round 6, call 2: the fixed step-loop finalize under the repro; A/B of it; A/B of the new scalar multiple; prover tests;
# window sweep of small plain MSMs; the self-calibrating bench line
tag=${1:-r6b}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; uptime) > $O/box.txt 2>&1
D=$PWD/distributed-groth16_amd
# 1. abort repro: the r5 form (a144/a255 clobber only) must abort, the same step loop with the registers named must not
for v in bsteps fsteps product; do
  l="DG16_X=0"; [ $v != product ] && l="DG16_LIB=$D/libdg16_$v.so"
  (env $l timeout 200 python -X faulthandler tools/repro_abort.py 25 2>&1 | grep -v "amdgpu.ids" | grep -E "iter|done|ERROR|MISMATCH|APERTURE|Abort" | tail -5; echo rc ${PIPESTATUS[0]}) > $O/repro_$v.txt
  echo "== repro $v"; tail -3 $O/repro_$v.txt
done
# 2. prover parity with the new scalar multiple
(timeout 500 python -X faulthandler -m pytest tests/test_gpu_prover.py tests/test_gpu_two_rank.py tests/test_gpu_hdist.py -q -m gpu -x 2>&1 | tail -5) > $O/pytest_prover.txt
cat $O/pytest_prover.txt
# 3. A/B: scalar multiple (prev = round 5's library), twice
for rep in 1 2; do
  for v in new prev; do
    l="DG16_X=0"; [ $v = prev ] && l="DG16_LIB=$D/libdg16_prev.so"
    echo "== $v" >> $O/ab_scalar_mul.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_scalar_mul.txt
    env $l timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab_scalar_mul.txt
  done
done
cat $O/ab_scalar_mul.txt
# 4. A/B: BLS12-381 2^20 proof, step-loop finalize (fixed) against the inlined one, twice
for rep in 1 2; do
  for v in product fsteps; do
    l="DG16_X=0"; [ $v = fsteps ] && l="DG16_LIB=$D/libdg16_fsteps.so"
    env $l timeout 200 python bench.py --curve bls12_381 --log-m 20 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bls12_381 2^20 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'g2 acc', round(d['roofline']['kernel_ms'],3))" >> $O/ab_fsteps.txt
  done
done
cat $O/ab_fsteps.txt
# 5. window sweep for small plain MSMs
for cg in "bn254 1" "bls12_377 1" "bn254 2"; do
  set -- $cg
  for c in 0 6 7 8 9 10 11 12; do
    e="DG16_X=0"; [ $c != 0 ] && e="DG16_MSM_C=$c"
    echo "c=$c" >> $O/msm_c_sweep.txt
    env $e timeout 100 python tools/msm_small_probe.py $1 $2 10,12,14,16 10 2>&1 | grep -v amdgpu.ids | sed 's/ms per call queued.*synchronised/ms q/' >> $O/msm_c_sweep.txt
  done
done
cat $O/msm_c_sweep.txt
# 6. the bench line with the in-run calibration
(timeout 300 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['single_proof_ms'], d['parity_check'], d['roofline']['kernel_ms'], d['calibration'])
print(d['valu_roofline']['frac'], d['valu_roofline_g1']['frac'])
PY
tail -3 $O/bench.err
