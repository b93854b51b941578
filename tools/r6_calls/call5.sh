# round 6, call 5: the king's combination on waves (d_msm, prove_a/b/c, packexp): dist tests + the dmsm sweep
tag=${1:-r6e}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 900 python -X faulthandler -m pytest tests/test_gpu_dist.py tests/test_gpu_two_rank.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest_dist.txt
cat $O/pytest_dist.txt
python - > $O/dmsm_sweep.json 2> $O/dmsm.err <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.dmsm_sweep(torch.device('cuda', 0), budget_s=60.0)))
PY
python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/dmsm_sweep.json').read().strip().splitlines()[-1])
for r in d['rows']:
    print(r['log_domain'], round(r['round_ms'], 2), 'ms', r['parity'], 'cpu clear', round(r['clear_msm_cpu_port_ms'], 1))
PY
tail -3 $O/dmsm.err
