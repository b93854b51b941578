O=gpurun_out/r6kk; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for n in 13 16; do
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/msm_small_probe.py bls12_377 1 $n 5 > $O/cmd$n.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_timeline.py "$db" 2.2 $O/timeline_bls12_377_g1_2e$n.md > /dev/null 2>&1
rm -rf $O/prof
done
