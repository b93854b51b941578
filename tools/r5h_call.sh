O=gpurun_out/r5h; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_hdist.py -m gpu -q -x 2>&1 | tail -5) > $O/gputest_ntt.txt
tail -2 $O/gputest_ntt.txt
for v in new old new old; do
  l=""; [ $v = old ] && l="DG16_LIB=/root/repo/gpurun_out_lib/libdg16_old.so"
  echo "== $v" >> $O/ab.txt
  for n in 21 22 20; do env $l python tools/perf_probe.py ntt $n 10 2>&1 | tail -1 >> $O/ab.txt; done
  env $l python tools/perf_probe.py hpoly 22 5 2>&1 | tail -1 >> $O/ab.txt
done
cat $O/ab.txt
