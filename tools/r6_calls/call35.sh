# round 6, call 35: the driver's other commands at the last library commit: the abort repro loop, smoke(), python bench.py
# with no flags
tag=${1:-r6ss}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 300 python -X faulthandler tools/repro_abort.py 30 2>&1 | tail -3; echo rc ${PIPESTATUS[0]}) > $O/repro.txt
cat $O/repro.txt
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
(time python bench.py) > $O/bench_default_flags.json 2> $O/bench_default_flags.err; tail -3 $O/bench_default_flags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6ss/bench_default_flags.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['parity_check'])
print([ (r['log_domain'], round(r['round_ms'],2)) for r in d['dmsm_sweep']['rows']])
PY
