# 8-shard per-rank timeline on one GPU + the BLS12-377 codec tests
O=gpurun_out/r3_s8; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 600 python -m pytest tests/test_arkkey.py -m gpu -q 2>&1 | tail -5) > $O/codec_test.txt
rm -rf $O/prof; rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/shard_timing.py 20 4 bn254 8 > $O/run.txt 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/stats_x8.md > /dev/null
python tools/rocprof_timeline.py "$db" 4.2 $O/timeline_x8.md > /dev/null 2>&1
rm -rf $O/prof
cat $O/codec_test.txt; tail -3 $O/run.txt
