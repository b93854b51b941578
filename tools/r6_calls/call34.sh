# round 6, call 34: window width of small BLS12-377 / BLS12-381 G1 MSMs (DG16_MSM_C) after the lane-form reductions
tag=${1:-r6rr}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for c in 0 7 8 9 10 11 12; do
  echo "== DG16_MSM_C=$c" >> $O/msm_c_sweep.txt
  if [ $c = 0 ]; then unset DG16_MSM_C; else export DG16_MSM_C=$c; fi
  timeout 200 python tools/msm_small_probe.py bls12_377 1 10,12,13,14,15,16 2>&1 | tail -6 >> $O/msm_c_sweep.txt
done
cat $O/msm_c_sweep.txt
