# round 6, call 20: timeline of a d_msm round (8 parties as threads on one GPU)
tag=${1:-r6bb}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/dmsm_probe.py 2 > $O/cmd.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_timeline.py "$db" 14 $O/timeline_dmsm.md > /dev/null 2>&1
python tools/rocprof_stats.py "$db" $O/stats_dmsm.md > /dev/null 2>&1
rm -rf $O/prof
cat $O/cmd.txt | tail -5
