# The evidence run of a round, ONE gpurun call (boxes of the pool differ by ~5 %: figures that are compared must come
# from one call):   gpurun --timeout 900 -- 'bash tools/evidence_run.sh <tag>'
#   gpurun_out/<tag>/gputest.txt                     full GPU suite, the driver's command
#   gpurun_out/<tag>/bench_line.json                 python bench.py --steps 20 --warmup 3
#   gpurun_out/<tag>/bench_prove_2e20_kernel_stats.md, timeline_one_proof.md
#                                                    rocprofv3 --kernel-trace --stats of the same bench (no extras)
# Copy what is to be judged into profiles/<round>_* afterwards (gpurun_out/ is scratch).
tag=${1:-evidence}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 420 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=300 2>&1 | tail -40) > $O/gputest.txt
tail -3 $O/gputest.txt
(timeout 240 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_line_profiled_run.json 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/bench_prove_2e20_kernel_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 14 $O/timeline_one_proof.md > /dev/null 2>&1
rm -rf $O/prof
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_check'], d['roofline']['kernel_ms'], d['g1_accumulate_ms'],
      d['valu_roofline']['frac'], d['valu_roofline']['whole_proof_valu_frac'])
PY
