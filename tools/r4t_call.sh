# Round 4: 48-byte G1 accumulation with the next point gathered ahead (pf48) against the shipped loop, same box.
O=gpurun_out/r4t; mkdir -p $O
L=distributed-groth16_amd
for pass in 1 2; do
  for v in base pf48; do
    lib=$L/libdg16_$v.so; [ $v = base ] && lib=$L/libdg16.so
    for c in bls12_381 bls12_377; do
      CURVE=$c DG16_LIB=$PWD/$lib timeout 100 python tools/perf_probe.py msm 20 5 2>&1 | tail -1 | sed "s/^/$v $c: /" >> $O/pf48_ab.txt
    done
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --curve bls12_381 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> $O/bench381_$v.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$v bls12_381 proof:', d['ms_per_step'], d['parity_check'])" >> $O/pf48_ab.txt
  done
done
cat $O/pf48_ab.txt
