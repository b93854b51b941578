# round 6, call 46: the atomic digit sort (small MSMs) with a thread per (scalar, window): timings, timeline at 2^10 / 2^13,
# the MSM / d_msm / prover GPU tests
tag=${1:-r6zv}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/small_sort_2d.txt
timeout 120 python tools/msm_small_probe.py bls12_377 1 10,11,12,13,14 10 2>&1 | grep "2^" >> $T
timeout 120 python tools/msm_small_probe.py bn254 1 10,12,13 10 2>&1 | grep "2^" >> $T
timeout 120 python tools/msm_small_probe.py bn254 2 10,12 10 2>&1 | grep "2^" >> $T
timeout 120 python tools/msm_small_probe.py bls12_381 2 10,12 10 2>&1 | grep "2^" >> $T
cat $T
for lg in 10 13; do
  out=$O/prof_$lg; rm -rf $out
  rocprofv3 --kernel-trace -d $out -o run -- python tools/msm_small_probe.py bls12_377 1 $lg 5 > /dev/null 2>&1
  db=$(find $out -name "*.db" | head -1)
  python tools/rocprof_timeline.py "$db" 1.2 $O/timeline_msm_bls12_377_g1_2e$lg.md > /dev/null 2>&1
  rm -rf $out
done
grep -E "digits|scatter" $O/timeline_msm_bls12_377_g1_2e1*.md | tail -4
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_dist.py tests/test_gpu_prover.py tests/test_gpu_zkey.py tests/test_gpu_qap.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cat $O/tests.txt
