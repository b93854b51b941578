# round 6, call 1: abort hunt + where a small plain MSM spends its time
tag=${1:-r6a}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
bash tools/abort_hunt.sh $tag 40
for cg in "bn254 1" "bls12_377 1" "bn254 2"; do
  set -- $cg
  timeout 120 python tools/msm_small_probe.py $1 $2 10,12,13,14,15,16 20 2>&1 | grep -v amdgpu.ids >> $O/msm_small.txt
done
cat $O/msm_small.txt
for cg in "bn254 1 12" "bls12_377 1 12" "bls12_377 1 15"; do
  set -- $cg
  rm -rf $O/prof
  timeout 120 rocprofv3 --kernel-trace -d $O/prof -o run -- python tools/msm_small_probe.py $1 $2 $3 3 > /dev/null 2> $O/prof.err
  db=$(find $O/prof -name "*.db" | head -1)
  python tools/rocprof_timeline.py "$db" 2.5 $O/timeline_msm_$1_g$2_2e$3.md > /dev/null 2>&1
done
rm -rf $O/prof
(timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline) > $O/bench_quick.json 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_proof_ms'), d['parity_check'], d['roofline']['kernel_ms'], d.get('g1_accumulate_ms'))
PY
