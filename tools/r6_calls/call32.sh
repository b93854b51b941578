# round 6, call 32: A, B1, L as three instances of ONE accumulation launch for short MSMs (DG16_ABL_MERGED=0 / 1), same library
tag=${1:-r6pp}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 900 python -X faulthandler -m pytest tests/test_gpu_prover.py tests/test_gpu_hdist.py tests/test_gpu_two_rank.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -4) > $O/pytest.txt
cat $O/pytest.txt
(DG16_ABL_MERGED=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_prover.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -4) > $O/pytest_forced.txt
cat $O/pytest_forced.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== DG16_ABL_MERGED=$v" >> $O/ab_abl_merged.txt
    DG16_ABL_MERGED=$v DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 2,4,8 2>&1 | grep "^world" >> $O/ab_abl_merged.txt
    DG16_ABL_MERGED=$v DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab_abl_merged.txt
    DG16_ABL_MERGED=$v timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_abl_merged.txt
  done
done
cat $O/ab_abl_merged.txt
