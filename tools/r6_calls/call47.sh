# round 6, call 47: plain-MSM window width (DG16_MSM_C) at 2^15 .. 2^19 points after the segment rule, BLS12-377 / BLS12-381 G1
tag=${1:-r6zu}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/msm_c_sweep_large.txt
for c in 0 11 12 13 14 15 16; do
  if [ $c = 0 ]; then unset DG16_MSM_C; else export DG16_MSM_C=$c; fi
  echo "DG16_MSM_C=$c" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_377 1 15,16,17,18,19 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 1 16,18,20 10 2>&1 | grep "2^" >> $T
done
cat $T
