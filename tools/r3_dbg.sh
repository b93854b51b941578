# does the order msm -> prover -> dist reproduce the stall seen once (last test never finished)?  faulthandler dumps the stacks
O=gpurun_out/r3_dbg; mkdir -p $O
(timeout 150 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline) > $O/bench_quick.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_dbg/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_check'], d['roofline']['kernel_ms'], d['valu_roofline']['frac'])
PY
(timeout 300 python -X faulthandler -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py -m gpu -q --durations=8 -o faulthandler_timeout=120 2>&1 | tail -80) > $O/subset.txt
tail -30 $O/subset.txt
