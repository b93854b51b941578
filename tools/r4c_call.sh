# Round 4, third GPU call: the GPU suite on the tree (wave-cooperative chains on the reduced-radix types, rows in chunks,
# out-of-line 14-limb G2 accumulation, fused Y3 for the 14-limb G1 loops), plain MSM timings + rocprof, and A/B against
# the library the session started from (libdg16_prentt.so = commit ddce44f) on both curves, with the NTT table knob.
O=gpurun_out/r4c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 800 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=300 2>&1 | tail -80) > $O/gputest.txt
tail -3 $O/gputest.txt
L=distributed-groth16_amd
for v in base prentt; do
  lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
  for what in "msm 20" "msm2 20" "msm 16" "msm 22"; do
    DG16_LIB=$PWD/$lib timeout 100 python tools/perf_probe.py $what 5 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/msm_ab.txt
  done
done
cat $O/msm_ab.txt
for pass in 1 2; do
  for v in base tw0 prentt; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so; [ $v = tw0 ] && lib=$L/libdg16.so
    tw=21; [ $v = tw0 ] && tw=0
    DG16_NTT_TABLE_MIN_LOG=$tw DG16_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench_$v.err | tail -1 >> $O/bench_$v.jsonl
  done
done
for pass in 1 2; do
  for v in base prentt; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --curve bls12_381 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> $O/bench381_$v.err | tail -1 >> $O/bench381_$v.jsonl
  done
done
bash tools/prof_run.sh r4c_msm_g1 "" python tools/perf_probe.py msm 20 5
bash tools/prof_run.sh r4c_msm_g2 "" python tools/perf_probe.py msm2 20 5
mv gpurun_out/r4c_msm_g1_* gpurun_out/r4c_msm_g2_* $O/ 2>/dev/null
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench*.jsonl')):
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith('{'): continue
        d = json.loads(ln)
        print(f.split('/')[-1], 'ms %.3f g2acc %.3f g1acc %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('g1_accumulate_ms', 0)))
PY
