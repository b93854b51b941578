# A/B inside one call: throughput G2 finalize (lanes per bucket 2 / 4 / 1) vs one lane per bucket on the side stream; h-poly on the side stream
O=gpurun_out/r3_fin; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -4) > $O/tests.txt
cat $O/tests.txt
run() { echo "== $*"; env "$@" timeout 200 python tools/perf_probe.py prove 20 10 2>/dev/null | grep groth16; }
(
run X=1
run DG16_FINALIZE_LDS=0
run DG16_FINALIZE_LPB=4
run DG16_FINALIZE_LPB=1
run DG16_HPOLY_SIDE=1
run DG16_HPOLY_SIDE=1 DG16_FINALIZE_LDS=0
run X=1
run DG16_FINALIZE_LDS=0
run CURVE=bls12_381
run CURVE=bls12_381 DG16_FINALIZE_LDS=0
run CURVE=bls12_381 DG16_FINALIZE_LPB=4
echo "== shard timing"; python tools/shard_timing.py 20 10 bn254 1,8 2>/dev/null | tail -3
echo "== shard timing, old finalize"; DG16_FINALIZE_LDS=0 python tools/shard_timing.py 20 10 bn254 8 2>/dev/null | tail -2
) > $O/ab.txt 2>&1
cat $O/ab.txt
