# round 6, call 45: dispatch timeline of an 8-shard rank's proofs in a queue at the last library commit
tag=${1:-r6zw}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
out=$O/prof; rm -rf $out
DG16_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $out -o run -- python tools/shard_timing.py 20 10 bn254 8 > $O/cmd.txt 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/shard8_kernel_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 6 $O/timeline_shard8.md > /dev/null 2>&1
rm -rf $out
grep "^world" $O/cmd.txt
