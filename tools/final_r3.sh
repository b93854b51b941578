# round-3 evidence run (on a gpurun box): tests, bench lines, rocprof stats + timeline, PMC traffic, shard timing
O=gpurun_out/r3_final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/gputest.txt
(timeout 600 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
rm -rf $O/prof; rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_line_profiled_run.json 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/bench_prove_2e20_kernel_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 14 $O/timeline_one_proof.md > /dev/null 2>&1
rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python tools/shard_timing.py 20 3 bn254 1 > /dev/null 2>&1
  python tools/pmc_sum.py $O/pmc_$c $c msm_accumulate_lds_kernel msm_accumulate_kernel ntt_step_kernel msm_finalize_thr_kernel msm_row_kernel >> $O/pmc_raw.txt
  rm -rf $O/pmc_$c
done
(timeout 300 python bench.py --steps 10 --warmup 2 --curve bls12_381 --no-extras) > $O/bench_line_bls12_381_2e20.json 2>> $O/bench.err
(timeout 300 python bench.py --steps 5 --warmup 2 --log-m 22 --no-extras) > $O/bench_line_bn254_2e22.json 2>> $O/bench.err
(timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --table-budget-gb 2.5) > $O/bench_line_budget_2p5GB.json 2>> $O/bench.err
python tools/shard_timing.py 20 10 bn254 1,2,4,8 2>/dev/null | tail -5 > $O/shard_timing.txt
DG16_ABL_MERGED=1 python tools/shard_timing.py 20 10 bn254 2,4,8 2>/dev/null | tail -4 >> $O/shard_timing.txt
(timeout 600 python bench.py --steps 3 --warmup 1 --curve bls12_381 --log-m 24 --no-extras) > $O/bench_line_bls12_381_2e24.json 2>> $O/bench.err
cat $O/gputest.txt; tail -c 400 $O/bench.err; cat $O/pmc_raw.txt; cat $O/shard_timing.txt
