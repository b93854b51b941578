# round-3 closing run: full GPU suite in the driver's order, the bench line, rocprof stats + timeline of the same bench
O=gpurun_out/r3_final2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 330 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=250 2>&1 | tail -40) > $O/gputest_full.txt
tail -3 $O/gputest_full.txt
(timeout 240 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
rm -rf $O/prof; timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_line_profiled_run.json 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/bench_prove_2e20_kernel_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 14 $O/timeline_one_proof.md > /dev/null 2>&1
rm -rf $O/prof
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_final2/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_check'], d['roofline']['kernel_ms'], d['roofline']['traffic_source'], d['valu_roofline']['frac'], d['valu_roofline']['whole_proof_valu_frac'])
PY
grep -i "finalize" $O/bench_prove_2e20_kernel_stats.md | cut -c1-200
