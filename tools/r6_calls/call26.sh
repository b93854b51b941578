# round 6, call 26: priority of the one-wave chain kernels (tail, scalar multiples, assembly: 3 against 0), same call
tag=${1:-r6jj}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
for rep in 1 2 3; do
  for v in prio3 prio0; do
    l="DG16_X=0"; [ $v = prio0 ] && l="DG16_LIB=$D/libdg16_chainprio0.so"
    echo "== $v" >> $O/ab_chain_prio.txt
    env $l DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 1,8 2>&1 | grep "^world" >> $O/ab_chain_prio.txt
    env $l DG16_OVERLAP=0 timeout 120 python tools/shard_timing.py 20 10 bn254 1,8 2>&1 | grep "^world" >> $O/ab_chain_prio.txt
    env $l timeout 120 python tools/config4_timing.py 2>&1 | tail -1 >> $O/ab_chain_prio.txt
  done
done
cat $O/ab_chain_prio.txt
