# round 6, call 48: c = 16 against the default for plain MSMs of 2^17 / 2^18 points, the other groups
tag=${1:-r6zt}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/msm_c16_other_groups.txt
for c in 0 16 0 16; do
  if [ $c = 0 ]; then unset DG16_MSM_C; else export DG16_MSM_C=$c; fi
  echo "DG16_MSM_C=$c" >> $T
  timeout 120 python tools/msm_small_probe.py bn254 1 17,18,19 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bn254 2 16,17,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 2 16,17,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_377 2 17,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 1 17,19 10 2>&1 | grep "2^" >> $T
done
cat $T
