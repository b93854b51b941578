# round 6, call 40: segment length of small PLAIN MSMs (DG16_MSM_SEG_LOG): the lane budget (kMinLanesLog) cuts the segments
# of a 2^16-point BLS12-377 G1 MSM to 8 entries, and the wave-per-bucket finalize then adds ~4 partials per bucket
tag=${1:-r6xx}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
T=$O/seg_log_small_plain_msm.txt
for sl in 0 3 4 5 6; do
  if [ $sl = 0 ]; then unset DG16_MSM_SEG_LOG; else export DG16_MSM_SEG_LOG=$sl; fi
  echo "seg_log=$sl" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_377 1 12,13,14,15,16,17,18,19 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 1 14,16,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bls12_381 2 14,16 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bn254 1 14,16,18 10 2>&1 | grep "2^" >> $T
  timeout 120 python tools/msm_small_probe.py bn254 2 14,16,18 10 2>&1 | grep "2^" >> $T
done
cat $T
