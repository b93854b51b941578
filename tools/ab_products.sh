# First measurement owed by round 3's last session (the GPU minutes of that round were spent when the field products
# became explicit v_mad_u64_u32 chains: profiles/r3c_static_products_as_instruction_chains.md).  ONE gpurun call:
#   (here, before the call)  bash tools/build_variant.sh noasm "-DDG29_NO_ASM_MAD -DDG29_NO_QUAD -DDG16_NO_POINT_PREFETCH"
#                             bash tools/build_variant.sh whole "-DDG29_ASM_WHOLE"      (every product ONE asm statement)
#   gpurun --timeout 900 -- 'bash tools/ab_products.sh'
# -> gpurun_out/ab_products/: the product rate of both forms (tools/ubench/fe_rate, compiled on the box), the bench line
#    of both libraries, kernel stats of both.  Then: refresh profiles/*_valu_constants.json from fe_rate_chain.json.
O=gpurun_out/ab_products; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fe_rate.hip -o /tmp/fe_rate_chain && /tmp/fe_rate_chain > $O/fe_rate_chain.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDG29_NO_ASM_MAD tools/ubench/fe_rate.hip -o /tmp/fe_rate_cxx && /tmp/fe_rate_cxx > $O/fe_rate_cxx.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDG29_ASM_WHOLE tools/ubench/fe_rate.hip -o /tmp/fe_rate_whole && /tmp/fe_rate_whole > $O/fe_rate_whole.json
for v in "" noasm whole "" noasm whole; do
  lib=distributed-groth16_amd/libdg16${v:+_$v}.so
  [ -f $lib ] || continue
  echo "== ${v:-chain}"
  DG16_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> $O/bench_${v:-chain}.err | tee -a $O/bench_${v:-chain}.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['g1_accumulate_ms'], d['parity_check'])"
done
for v in "" noasm whole; do
  lib=distributed-groth16_amd/libdg16${v:+_$v}.so
  [ -f $lib ] || continue
  rm -rf $O/prof
  DG16_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2> $O/prof_${v:-chain}.err
  db=$(find $O/prof -name "*.db" | head -1)
  python tools/rocprof_stats.py "$db" $O/kernel_stats_${v:-chain}.md > /dev/null
  rm -rf $O/prof
done
