# Round 4, fifth GPU call: GPU suite; the pipelined plain MSM against DG16_MSM_PIPELINE=0 (same library); the 14-limb G2
# accumulation in lockstep (ls128 / ls256) against the tree on BLS12-381; BN254 proof base vs the session's start.
O=gpurun_out/r4e; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 800 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=300 2>&1 | tail -80) > $O/gputest.txt
tail -3 $O/gputest.txt
L=distributed-groth16_amd
for pl in 1 0; do
  for what in "msm 20" "msm2 20" "msm 22" "msm 18"; do
    DG16_MSM_PIPELINE=$pl timeout 100 python tools/perf_probe.py $what 5 2>&1 | tail -1 | sed "s/^/pipeline=$pl bn254: /" >> $O/msm_ab.txt
  done
  for what in "msm 20" "msm2 20"; do
    CURVE=bls12_381 DG16_MSM_PIPELINE=$pl timeout 100 python tools/perf_probe.py $what 5 2>&1 | tail -1 | sed "s/^/pipeline=$pl bls12_381: /" >> $O/msm_ab.txt
  done
done
cat $O/msm_ab.txt
for pass in 1 2; do
  for v in base ls128 ls256; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --curve bls12_381 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> $O/bench381_$v.err | tail -1 >> $O/bench381_$v.jsonl
  done
done
for pass in 1 2; do
  for v in base prentt; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench_$v.err | tail -1 >> $O/bench_$v.jsonl
  done
done
bash tools/prof_run.sh r4e_msm_g1 "" python tools/perf_probe.py msm 20 5
bash tools/prof_run.sh r4e_msm_g2 "" python tools/perf_probe.py msm2 20 5
mv gpurun_out/r4e_msm_* $O/ 2>/dev/null
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench*.jsonl')):
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith('{'): continue
        d = json.loads(ln)
        print(f.split('/')[-1], 'ms %.3f g2acc %.3f g1acc %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('g1_accumulate_ms', 0)))
PY
