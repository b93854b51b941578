# A/B of variant builds (tools/build_variant_fast.sh NAME ...) against the tree's library in ONE gpurun call:
#   gpurun -- 'bash tools/ab_variants.sh <tag> base g2b128 ntt3 seg8'        -> gpurun_out/<tag>/ab.txt
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
for rep in 1 2 3; do
  for v in "$@"; do
    l=""; [ $v != base ] && l="DG16_LIB=$PWD/distributed-groth16_amd/libdg16_$v.so"
    env $l python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'G2 acc', round(d['roofline']['kernel_ms'],3), 'g1 acc', round(d['g1_accumulate_ms'],3))" >> $O/ab.txt
  done
done
cat $O/ab.txt
