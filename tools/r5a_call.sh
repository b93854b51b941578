O=gpurun_out/r5a; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_hdist.py -m gpu -q -k "config5" --durations=5 2>&1 | tail -15) > $O/gputest_config5.txt
(timeout 200 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline) > $O/bench_line_head.json 2> $O/bench.err
(timeout 300 python bench.py --curve bls12_381 --log-m 20 --shards-in-process 8 --steps 3) > $O/bench_shards_bls_2e20.json 2>> $O/bench.err
bash tools/sq_evidence.sh r5a
tail -5 $O/gputest_config5.txt; cat $O/passes.txt
