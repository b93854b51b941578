# Round 4: window width of the resident tables once more, now that the last reduction is off the critical path.
O=gpurun_out/r4w; mkdir -p $O
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$lbl', 'ms_per_step', round(d['ms_per_step'], 3), 'single', round(d['single_proof_ms'], 3), 'g2acc', round(d['roofline']['kernel_ms'], 3), 'g1acc', round(d['g1_accumulate_ms'], 3), d['config']['key_window_bits'])" >> $O/window_ab.txt
}
run base X=1
run c_h=19 DG16_MSM_TABLE_C_H=19
run c_h=20 DG16_MSM_TABLE_C_H=20
run c=19 DG16_MSM_TABLE_C=19
run c=20 DG16_MSM_TABLE_C=20
run c=18 DG16_MSM_TABLE_C=18
run base X=1
cat $O/window_ab.txt; tail -3 $O/bench.err
