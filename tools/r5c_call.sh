# full GPU suite at HEAD + A/B of the 14-limb G2 accumulation forms + clock / issue-rate calibration
O=gpurun_out/r5c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -o faulthandler_timeout=400 --durations=8 2>&1 | tail -25) > $O/gputest.txt
tail -3 $O/gputest.txt
for v in inline steps inline steps; do
  e=""; [ $v = inline ] && e="DG16_G2_14LIMB=inline"
  echo "== $v" >> $O/ab.txt
  env $e CURVE=bls12_381 python tools/perf_probe.py msm2 20 5 2>&1 | tail -1 >> $O/ab.txt
  env $e python bench.py --curve bls12_381 --log-m 20 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('proof ms', d['ms_per_step'], 'single', d['single_proof_ms'], 'G2 acc ms', d['roofline']['kernel_ms'], 'g1 acc', d['g1_accumulate_ms'])" >> $O/ab.txt
done
cat $O/ab.txt
# clock under the accumulation kernels: GRBM_GUI_ACTIVE (cycles) / kernel duration
rm -rf $O/pmc_clk
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_clk -o run -- python tools/shard_timing.py 20 2 bn254 1 > $O/pmc_clk.log 2>&1
python - "$O" <<'PY' > $O/clock_under_kernels.txt 2>&1
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + "/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        k = r["Kernel_Name"].split("(")[0][:90]
        a = acc[k]
        a[0] += float(r["Counter_Value"]); a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[2] += 1
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda x: -x[1][1])[:14]:
    print("%-92s launches %3d  avg %8.1f us  GRBM_GUI_ACTIVE/launch %.3e  -> %.0f MHz (if summed over %s: see raw)" % (k, n, ns / n / 1e3, cyc / n, cyc / ns * 1e3, "1 instance"))
PY
rm -rf $O/pmc_clk
make -s -C tools/ubench instr_rate > /dev/null 2>&1 && (cd tools/ubench && timeout 120 ./instr_rate 2>&1 | grep -E "device|k_mad_u64_u32|k_and_or|k_add_u32|k_mul_lo|k_lshl_add" ) > $O/ubench_instr_rate.txt
cat $O/clock_under_kernels.txt | head -8; cat $O/ubench_instr_rate.txt
