# round 6, call 43: config 5's size at the round's last library: BLS12-381 2^24 on ONE GPU (timed, the oracle proves the
# timed instance) and as eight shard keys in one process against the oracle; then the driver's other commands
tag=${1:-r6end}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 900 python bench.py --curve bls12_381 --log-m 24 --steps 3 --warmup 1 --full-parity --no-extras) > $O/bench_line_bls12_381_2e24_full_parity.json 2> $O/bench_2e24.err
tail -c 600 $O/bench_line_bls12_381_2e24_full_parity.json; tail -3 $O/bench_2e24.err
(timeout 900 python bench.py --curve bls12_381 --log-m 24 --shards-in-process 8 --steps 2 --full-parity) > $O/bench_shards_in_process_bls12_381_2e24_full_parity.json 2>> $O/bench_2e24.err
tail -c 600 $O/bench_shards_in_process_bls12_381_2e24_full_parity.json; tail -3 $O/bench_2e24.err
(timeout 300 python -X faulthandler tools/repro_abort.py 30 2>&1 | tail -3; echo rc ${PIPESTATUS[0]}) > $O/repro.txt
cat $O/repro.txt
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
(time python bench.py) > $O/bench_default_flags.json 2> $O/bench_default_flags.err; tail -3 $O/bench_default_flags.err
