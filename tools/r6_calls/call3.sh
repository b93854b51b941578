# round 6, call 3: calibration cross-check; scalar multiple A/B with dense r, s; stage-1 kernel times; full GPU suite
tag=${1:-r6c}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; uptime) > $O/box.txt 2>&1
D=$PWD/distributed-groth16_amd
# 1. calibration: the library's figure at several lengths next to tools/ubench/instr_rate in the same call
python - > $O/calib_check.txt 2>&1 <<'PY'
import ctypes, os
lib = ctypes.CDLL(os.path.join("distributed-groth16_amd", "libdg16_calib.so"))
lib.dg16_calib_mad_rate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for rep in range(2):
    for wps in (8, 4, 2, 1):
        for iters in (1024, 4096, 16384, 65536):
            buf = (ctypes.c_double * 4)()
            rc = lib.dg16_calib_mad_rate(0, wps, iters, buf)
            print("wps %d iters %6d rc %d: %.2f T lane-op/s, sclk %.0f MHz, %.3f ms" % (wps, iters, rc, buf[0], buf[1], buf[2]), flush=True)
PY
(timeout 60 tools/ubench/instr_rate 2>&1 | grep -E "device|mad_u64|and_or|add_u32 ") >> $O/calib_check.txt
cat $O/calib_check.txt
# 2. scalar multiple A/B with dense r, s (bench.Workload since this commit), twice
for rep in 1 2; do
  for v in new prev; do
    l="DG16_X=0"; [ $v = prev ] && l="DG16_LIB=$D/libdg16_prev.so"
    echo "== $v" >> $O/ab_scalar_mul.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_scalar_mul.txt
    env $l timeout 120 python tools/shard_timing.py 20 10 bn254 8 2>&1 | grep "^world" >> $O/ab_scalar_mul.txt
  done
done
cat $O/ab_scalar_mul.txt
# 3. the stage-1 kernel itself under rocprofv3 (8-way shard), both libraries
for v in new prev; do
  l="DG16_X=0"; [ $v = prev ] && l="DG16_LIB=$D/libdg16_prev.so"
  rm -rf $O/prof
  env $l timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/shard_timing.py 20 10 bn254 8 > /dev/null 2> $O/prof.err
  db=$(find $O/prof -name "*.db" | head -1)
  python tools/rocprof_stats.py "$db" $O/shard8_kernel_stats_$v.md > /dev/null
  grep -E "stage1|assemble|msm_tail|msm_top|msm_row" $O/shard8_kernel_stats_$v.md | cut -c1-200
done
rm -rf $O/prof
# 4. small plain MSMs with the window rule
for cg in "bn254 1" "bn254 2"; do
  set -- $cg
  timeout 100 python tools/msm_small_probe.py $1 $2 10,12,13,14 10 2>&1 | grep -v amdgpu.ids >> $O/msm_small.txt
done
cat $O/msm_small.txt
# 5. the GPU suite
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=400 > $O/gputest_full.txt 2>&1
(grep -E "passed|failed|error|Fatal|File \"/root/repo|File \"/tmp/code" $O/gputest_full.txt | tail -30; grep -E "^tests/|^\.+|^=+|^FAILED|^ERROR" $O/gputest_full.txt | tail -14) > $O/gputest.txt
tail -6 $O/gputest.txt
