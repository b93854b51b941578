# round 6, call 6: three bounded experiments on the G1 accumulation's tree (same call, A/B x 2); task-parallel placement
# probe; config-4 timeline
tag=${1:-r6f}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
D=$PWD/distributed-groth16_amd
for rep in 1 2; do
  for v in product g1outline g1into g1b128; do
    l="DG16_X=0"; [ $v != product ] && l="DG16_LIB=$D/libdg16_$v.so"
    env $l timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'proof ms', round(d['ms_per_step'],3), 'single', round(d['single_proof_ms'],3), 'A acc ms', round(d['g1_accumulate_ms'],4), 'G2 acc', round(d['roofline']['kernel_ms'],3), d['parity_check'])" >> $O/g1_tree_experiments.txt
  done
done
cat $O/g1_tree_experiments.txt
# task-parallel placement ("the G2 MSM can take 2-3 GPUs alone"): what the slowest rank of {B on 3 ranks, A / B1 / L / H
# on 5} would run, timed alone on this GPU through resident tables
python - > $O/task_parallel_probe.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import dg16_amd
ctx = dg16_amd.Context(0)
dev = torch.device('cuda:0')
def resident(group, n, reps=10):
    pb = 64 * group
    bases = torch.empty(n * pb, dtype=torch.uint8, device=dev)
    ctx.gen_bases_dev('bn254', group, 5, n, bases.data_ptr()); ctx.sync(0)
    lo = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev)
    hi = torch.randint(0, 0x30644E72E131A029, (n, 1), dtype=torch.int64, device=dev)
    sc = torch.cat([lo, hi], dim=1).contiguous()
    out = torch.empty(3 * 32 * group, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hb = ctx.bases_upload('bn254', group, bases.data_ptr(), n, device_ptrs=True)
    for _ in range(3):
        ctx.msm_resident_dev(hb, sc.data_ptr(), n, out.data_ptr()); 
    ctx.sync(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_resident_dev(hb, sc.data_ptr(), n, out.data_ptr()); ctx.sync(0)
    ms = (time.perf_counter() - t0) / reps * 1e3
    acc = ctx.last_kernel_ms(0, 1)
    hb.close()
    return ms, acc
n = 1 << 20
for name, group, cnt in (("G2 third (B on 3 ranks)", 2, n // 3 + 1), ("G2 half (B on 2 ranks)", 2, n // 2),
                         ("G1 fifth (one of A/B1/L/H on 5 ranks)", 1, n // 5 + 1), ("G1 eighth (uniform shard)", 1, n // 8),
                         ("G2 eighth (uniform shard)", 2, n // 8)):
    ms, acc = resident(group, cnt)
    print("%-42s n = %7d: whole resident MSM %.3f ms (sort + accumulation %.3f ms + reduction chain)" % (name, cnt, ms, acc), flush=True)
PY
cat $O/task_parallel_probe.txt | grep -v amdgpu.ids
# where a config-4 proof spends its time now
rm -rf $O/prof
timeout 150 rocprofv3 --kernel-trace -d $O/prof -o run -- python tools/config4_timing.py > /dev/null 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_timeline.py "$db" 2.2 $O/timeline_config4.md > /dev/null 2>&1
rm -rf $O/prof
awk -F'|' 'NR>2{printf "%8s %8s %3s %s\n", $2, $3, $4, substr($5,1,70)}' $O/timeline_config4.md | tail -50
