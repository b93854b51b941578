# A/B of library builds inside ONE gpurun call (boxes of the pool differ by ~5 %).  Builds are made here, before the
# call, with   bash tools/build_variant.sh <name> "<flags>"   -> distributed-groth16_amd/libdg16_<name>.so
# and every libdg16_*.so present is timed next to the tree's libdg16.so ("base"), twice, interleaved:
#   gpurun --timeout 900 -- 'bash tools/ab_products.sh [steps]'
# -> gpurun_out/ab_products/: bench_<name>.jsonl (the bench lines), summary.txt (ms per proof, G2 / G1 accumulation ms,
#    parity), fe_rate_chain.json (tools/ubench/fe_rate on the shipped product form: profiles/*_valu_constants.json).
steps=${1:-20}
O=gpurun_out/ab_products; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fe_rate.hip -o /tmp/fe_rate_chain && /tmp/fe_rate_chain > $O/fe_rate_chain.json
names="base"
for f in distributed-groth16_amd/libdg16_*.so; do
  [ -f "$f" ] || continue
  v=${f##*/libdg16_}; names="$names ${v%.so}"
done
: > $O/summary.txt
for pass in 1 2; do
  for v in $names; do
    lib=distributed-groth16_amd/libdg16_$v.so; [ $v = base ] && lib=distributed-groth16_amd/libdg16.so
    DG16_LIB=$PWD/$lib timeout 200 python bench.py --steps $steps --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench_$v.err | tee -a $O/bench_$v.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pass $pass: ms_per_proof %.3f g2_acc_ms %.3f g1_acc_ms %.3f parity %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['g1_accumulate_ms'], d['parity_check']))" | tee -a $O/summary.txt
  done
done
