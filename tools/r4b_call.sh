# Round 4, second GPU call: the GPU suite on the tree (direct NTT twiddle tables), the l = 1 MPC-party check with its full
# output, A/B of the pending build variants (BN254: prentt = HEAD without the twiddle tables, g1prefetch, noasm;
# BLS12-381: fuseall, outl381), stand-alone NTT / h-polynomial timings of base vs prentt, rocprof of the plain MSMs.
O=gpurun_out/r4b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 700 python -X faulthandler -m pytest tests -m gpu -q -x -o faulthandler_timeout=300 2>&1 | tail -60) > $O/gputest.txt
tail -3 $O/gputest.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29581 tests/mpc_rank_check.py 1 6 40 > $O/mpc_l1.txt 2>&1
echo "mpc l=1 rc $?"; grep -n "Error\|error\|PASS\|FAIL\|assert" $O/mpc_l1.txt | head -20
L=distributed-groth16_amd
for pass in 1 2; do
  for v in base prentt g1prefetch noasm; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    [ -f $lib ] || continue
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench_$v.err | tail -1 >> $O/bench_$v.jsonl
  done
done
for pass in 1 2; do
  for v in base fuseall outl381; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    [ -f $lib ] || continue
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --curve bls12_381 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> $O/bench381_$v.err | tail -1 >> $O/bench381_$v.jsonl
  done
done
for v in base prentt; do
  lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
  for what in "ntt 22" "ntt 20" "ntt 24" "hpoly 20"; do
    DG16_LIB=$PWD/$lib timeout 100 python tools/perf_probe.py $what 10 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/ntt_ab.txt
  done
done
cat $O/ntt_ab.txt
bash tools/prof_run.sh r4b_msm_g1 "" python tools/perf_probe.py msm 20 5
bash tools/prof_run.sh r4b_msm_g2 "" python tools/perf_probe.py msm2 20 5
mv gpurun_out/r4b_msm_g1_* gpurun_out/r4b_msm_g2_* $O/ 2>/dev/null
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench*.jsonl')):
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith('{'): continue
        d = json.loads(ln)
        print(f.split('/')[-1], 'ms %.3f g2acc %.3f g1acc %.3f parity %s ntt22 %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('g1_accumulate_ms', 0), d.get('parity_check'), (d.get('ntt_2^22') or {}).get('ms')))
PY
