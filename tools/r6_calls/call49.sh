# round 6, call 49: c = 16 for plain G1 MSMs of 2^17 .. 2^18 points as the default: timings and the MSM / d_msm GPU tests
tag=${1:-r6zs}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/plain_g1_c16_default.txt
timeout 120 python tools/msm_small_probe.py bls12_377 1 16,17,18,19 10 2>&1 | grep "2^" >> $T
timeout 120 python tools/msm_small_probe.py bn254 1 16,17,18,19 10 2>&1 | grep "2^" >> $T
timeout 120 python tools/msm_small_probe.py bls12_381 1 16,17,18,19 10 2>&1 | grep "2^" >> $T
cat $T
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cat $O/tests.txt
