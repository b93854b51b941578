# round 6, call 21: inversion / affine form in lane form (d_msm's two inversions per round): probe, dist + msm parity,
# d_msm sweep
tag=${1:-r6cc}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout 200 tools/ubench/lane29_probe > $O/lane29_probe.txt 2>&1; head -8 $O/lane29_probe.txt
(timeout 1000 python -X faulthandler -m pytest tests/test_gpu_dist.py tests/test_gpu_msm.py tests/test_gpu_lane29.py tests/test_gpu_two_rank.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert|^tests" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python tools/dmsm_probe.py 40 > $O/dmsm_sweep.txt 2>&1; tail -12 $O/dmsm_sweep.txt
