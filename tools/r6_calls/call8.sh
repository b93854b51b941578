# round 6, call 8: s*A' / r*B1' on two waves per chain: prover parity, config 4 and 8-shard A/B against the one-wave form
tag=${1:-r6h}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
(timeout 600 python -X faulthandler -m pytest tests/test_gpu_prover.py tests/test_gpu_hdist.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -6) > $O/pytest_prover.txt
cat $O/pytest_prover.txt
for rep in 1 2; do
  for v in twowaves onewave; do
    l="DG16_X=0"; [ $v = onewave ] && l="DG16_LIB=$D/libdg16_onewave.so"
    echo "== $v" >> $O/ab_stage1_two_waves.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_stage1_two_waves.txt
    env $l DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab_stage1_two_waves.txt
  done
done
cat $O/ab_stage1_two_waves.txt
