O=gpurun_out/r6x; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/config4_timing.py > $O/cmd.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_timeline.py "$db" 0.9 $O/timeline_config4.md > /dev/null 2>&1
rm -rf $O/prof
