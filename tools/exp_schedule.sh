# Schedule experiments of the single-GPU prover in ONE gpurun call (env knobs of csrc/prover_impl.h: DG16_EXP bits,
# DG16_ABL_MERGED; the two-lane form needs DG16_MAIN2=1 and a fifth hardware queue: GPU_MAX_HW_QUEUES).
#   gpurun --timeout 900 -- 'bash tools/exp_schedule.sh [steps]'   -> gpurun_out/exp_schedule/summary.txt
steps=${1:-20}
O=gpurun_out/exp_schedule; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
: > $O/summary.txt
run() {   # name, env assignments...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps $steps --warmup 3 --no-extras --no-cpu-baseline 2> $O/$name.err | tee -a $O/$name.jsonl | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t and t[-1].startswith('{') else None
print('$name [$*]: ' + ('ms_per_proof %.3f g2_acc %.3f g1_acc %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['g1_accumulate_ms']) if d else 'FAILED'))" | tee -a $O/summary.txt
}
for pass in 1 2; do
  run base X=0
  run merged DG16_ABL_MERGED=1
  run b_red_last DG16_EXP=4
  run merged_b_red_last DG16_ABL_MERGED=1 DG16_EXP=4
  run no_b_red DG16_EXP=1
  run no_side_chains DG16_EXP=3
  run hwq8 GPU_MAX_HW_QUEUES=8
  run two_lane GPU_MAX_HW_QUEUES=8 DG16_MAIN2=1 DG16_EXP=16
  run two_lane_b_red_last GPU_MAX_HW_QUEUES=8 DG16_MAIN2=1 DG16_EXP=20
done
