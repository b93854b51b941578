O=gpurun_out/r5g; mkdir -p $O
(timeout 200 python tools/shard_timing.py 20 10 bn254 1,2,4,8 2>&1 | grep -E "^world|per_rank") > $O/shard_timing_a.txt
(timeout 200 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep -E "^world|per_rank") > $O/shard_timing_b.txt
cat $O/shard_timing_a.txt $O/shard_timing_b.txt
(timeout 900 python bench.py --curve bls12_381 --log-m 24 --shards-in-process 8 --steps 2 --full-parity) > $O/bench_shards_in_process_bls12_381_2e24_full_parity.json 2> $O/err.txt
tail -c 700 $O/bench_shards_in_process_bls12_381_2e24_full_parity.json
(timeout 400 python bench.py --curve bls12_381 --log-m 22 --shards-in-process 8 --steps 2) > $O/bench_shards_in_process_bls12_381_2e22.json 2>> $O/err.txt
(timeout 600 python bench.py --curve bls12_381 --log-m 24 --steps 3 --warmup 1 --no-extras --no-cpu-baseline) > $O/bench_line_bls12_381_2e24.json 2>> $O/err.txt
python -c "
import json
d=json.loads(open('$O/bench_line_bls12_381_2e24.json').read().strip().splitlines()[-1]); print('2^24 single GPU', d['ms_per_step'], d['roofline']['kernel_ms'], d['g1_accumulate_ms'])"
tail -3 $O/err.txt
