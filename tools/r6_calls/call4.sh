# round 6, call 4: overlapped queue of sharded proofs (tests + per-rank timing), the bench line with its new extras
tag=${1:-r6d}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; uptime) > $O/box.txt 2>&1
(timeout 600 python -X faulthandler -m pytest tests/test_gpu_hdist.py tests/test_gpu_two_rank.py -q -m gpu -x -k "overlapped or sharded_proof_across or bench_flow" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8) > $O/pytest_overlap.txt
cat $O/pytest_overlap.txt
for rep in 1 2; do
  for ov in 0 1; do
    env DG16_OVERLAP=$ov timeout 200 python tools/shard_timing.py 20 10 bn254 1,2,4,8 2>&1 | grep "^world" >> $O/shard_timing_overlap_ab.txt
  done
done
cat $O/shard_timing_overlap_ab.txt
(time timeout 600 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
tail -4 $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['single_proof_ms'], d['parity_check'], d['roofline']['kernel_ms'])
print('calib', d['calibration'])
print('valu', d['valu_roofline']['frac'], d['valu_roofline_g1']['frac'], d['valu_roofline']['whole_proof_valu_frac'])
print('ttfp', d['config'].get('time_to_first_proof_s'), d['config'].get('key_table_build_s'))
print('tableless', d.get('tableless_proof'))
print('dmsm', json.dumps(d.get('dmsm_sweep'))[:1500])
print('sweep', [(r['log_n'], round(r['gpu_ms'], 3)) for r in d['msm_sweep']['rows']])
print('c4', d.get('config4_sha256_shaped'))
print('c5', {k: v for k, v in d.get('config5_bls12_381_2e20', {}).items() if k in ('ms_per_proof', 'parity_check', 'roofline')})
PY
