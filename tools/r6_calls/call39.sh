# round 6, call 39: the final top-window rule (BLS12-381 keeps lg + 1): table tests; timelines of small plain MSMs
# (BLS12-377 G1: the msm_bench.rs shape) and of config 4 at the new width
tag=${1:-r6ww}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_hdist.py tests/test_gpu_dist.py tests/test_gpu_two_rank.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cat $O/tests.txt
timeout 120 python tools/config4_timing.py 2>&1 | tail -1 > $O/config4_timing.txt
DG16_OVERLAP=1 timeout 160 python tools/shard_timing.py 16 5 bls12_381 4,8 2>&1 | grep "^world" > $O/shard_bls12_381_2e16.txt
for lg in 10 13 16; do
  out=$O/prof_$lg; rm -rf $out
  rocprofv3 --kernel-trace -d $out -o run -- python tools/msm_small_probe.py bls12_377 1 $lg 5 > $O/msm_bls12_377_g1_2e$lg.txt 2>&1
  db=$(find $out -name "*.db" | head -1)
  python tools/rocprof_timeline.py "$db" 2.2 $O/timeline_msm_bls12_377_g1_2e$lg.md > /dev/null 2>&1
  rm -rf $out
done
out=$O/prof_c4; rm -rf $out
rocprofv3 --kernel-trace -d $out -o run -- python tools/config4_timing.py > /dev/null 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/rocprof_timeline.py "$db" 3.2 $O/timeline_config4.md > /dev/null 2>&1
rm -rf $out
cat $O/config4_timing.txt $O/shard_bls12_381_2e16.txt $O/msm_bls12_377_g1_2e*.txt
