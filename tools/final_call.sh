# One gpurun call: box identity, the abort repro loop, the proof assembly A/B (previous library against the tree, DG16_LIB),
# then the driver's GPU suite and bench line on the tree.   gpurun --timeout 900 -- 'bash tools/final_call.sh <tag>'
tag=${1:-final}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i "unique\|serial"; uptime) > $O/box.txt 2>&1
P=$PWD/distributed-groth16_amd/libdg16_prev.so
if [ -f $P ]; then
  for rep in 1 2; do
    for v in new prev; do
      l=""; [ $v = prev ] && l="DG16_LIB=$P"
      echo "== $v" >> $O/ab.txt
      env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab.txt
      env $l timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab.txt
    done
  done
  for v in new prev; do
    l=""; [ $v = prev ] && l="DG16_LIB=$P"
    env $l timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bn254 proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), d['parity_check'])" >> $O/ab.txt
  done
  cat $O/ab.txt
fi
(timeout 200 python -X faulthandler tools/repro_abort.py 30 2>&1 | tail -3; echo rc ${PIPESTATUS[0]}) > $O/repro.txt
cat $O/repro.txt
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=400 > $O/gputest_full.txt 2>&1
(grep -E "passed|failed|error|Fatal|File \"/root/repo|File \"/tmp/code" $O/gputest_full.txt | tail -30; grep -E "^tests/|^\.+|^=+" $O/gputest_full.txt | tail -12) > $O/gputest.txt
tail -3 $O/gputest.txt
(timeout 300 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['single_proof_ms'], d['parity_check'], d['roofline']['kernel_ms'])
PY
