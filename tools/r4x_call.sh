# Round 4: R1CS x witness of the next proof on channel 1 (under the last proof's H accumulation) -- experiment
O=gpurun_out/r4x; mkdir -p $O
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --cpu-sample-log 16 2> $O/bench.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$lbl', 'ms_per_step', round(d['ms_per_step'], 3), 'single', round(d['single_proof_ms'], 3), d.get('parity_check'))" >> $O/qap_side_ab.txt
}
run base X=1
run qap_ch1 DG16_EXP=64 DG16_BENCH_QAP_CH=1
run base X=1
run qap_ch1 DG16_EXP=64 DG16_BENCH_QAP_CH=1
cat $O/qap_side_ab.txt; tail -3 $O/bench.err
