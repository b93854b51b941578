# round 6, call 25: priority of the row / top / 16-wave reduction kernels (3 against 0) in a queue of proofs, same call
tag=${1:-r6ii}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
for rep in 1 2 3; do
  for v in prio3 prio0; do
    l="DG16_X=0"; [ $v = prio0 ] && l="DG16_LIB=$D/libdg16_redprio0.so"
    echo "== $v" >> $O/ab_reduce_prio.txt
    env $l DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 1,2 2>&1 | grep "^world" >> $O/ab_reduce_prio.txt
    env $l DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 1 2>&1 | grep "^world" >> $O/ab_reduce_prio.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_reduce_prio.txt
  done
done
cat $O/ab_reduce_prio.txt
