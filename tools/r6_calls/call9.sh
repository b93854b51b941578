tag=${1:-r6i}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
(timeout 300 python -X faulthandler -m pytest tests/test_gpu_prover.py -q -m gpu -x -k "medium_sizes" 2>&1 | tail -3) > $O/pytest.txt; cat $O/pytest.txt
for rep in 1 2 3; do
(timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['g1_accumulate_ms'],3), 'calib', round(d['calibration']['mad_issue_T_lane_ops_per_s'],2), round(d['calibration']['sclk_under_mad_load_mhz']), 'frac', round(d['valu_roofline']['frac'],3), round(d['valu_roofline_g1']['frac'],3))
print('  G2', d['valu_roofline']['cycle_view']); print('  G1', d['valu_roofline_g1']['cycle_view'])" >> $O/cycle_view.txt
done
cat $O/cycle_view.txt
