# The evidence run of a round, ONE gpurun call (boxes of the pool differ by ~5 %: figures that are compared must come
# from one call):   gpurun --timeout 1200 -- 'bash tools/evidence_run.sh <tag> [pytest selection]'
#   gpurun_out/<tag>/gputest.txt                     GPU suite (the driver's command; or the given selection)
#   gpurun_out/<tag>/bench_line.json                 python bench.py --steps 20 --warmup 3
#   gpurun_out/<tag>/bench_prove_2e20_kernel_stats.md, timeline_one_proof.md
#                                                    rocprofv3 --kernel-trace --stats of the same bench (no extras)
#   gpurun_out/<tag>/bench_line_bls12_381_2e20.json, bench_shards_in_process_bls12_381_2e20.json, shard_timing.txt
#   gpurun_out/<tag>/pmc_raw.txt, pmc_g2_accumulate.json
#                                                    separate --pmc FETCH_SIZE / WRITE_SIZE passes over three proofs
#                                                    (tools/pmc_sum.py): what bench.py reads roofline.traffic from
# Copy what is to be judged into profiles/<round>_* afterwards (gpurun_out/ is scratch).
tag=${1:-evidence}; shift
sel=${*:-tests}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
git_head=$(cat .git_head 2>/dev/null || echo unknown)
timeout 900 python -X faulthandler -m pytest $sel -m gpu -q -o faulthandler_timeout=400 > $O/gputest_full.txt 2>&1
(grep -E "passed|failed|error|Fatal|File \"/root/repo|File \"/tmp/code" $O/gputest_full.txt | tail -30; grep -E "^tests/|^\.+|^=+" $O/gputest_full.txt | tail -12) > $O/gputest.txt
tail -3 $O/gputest.txt
(timeout 300 python bench.py --steps 20 --warmup 3) > $O/bench_line.json 2> $O/bench.err
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_line_profiled_run.json 2> $O/prof.err
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" $O/bench_prove_2e20_kernel_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 14 $O/timeline_one_proof.md > /dev/null 2>&1
rm -rf $O/prof
: > $O/pmc_raw.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc -o run -- python tools/shard_timing.py 20 3 bn254 1 > $O/pmc_$ctr.log 2>&1
  python tools/pmc_sum.py $O/pmc $ctr msm_accumulate_lds_kernel msm_accumulate_kernel ntt_step_kernel msm_finalize msm_row_kernel msm_top_kernel qap_kernel >> $O/pmc_raw.txt
  rm -rf $O/pmc
done
python tools/pmc_json.py $O/pmc_raw.txt "$tag" > $O/pmc_g2_accumulate.json
# ... and the same two passes over BLS12-381 proofs (config 5's curve: the 14-limb kernels)
: > $O/pmc_raw_bls12_381.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc -o run -- python tools/shard_timing.py 20 3 bls12_381 1 > $O/pmc_bls_$ctr.log 2>&1
  python tools/pmc_sum.py $O/pmc $ctr msm_accumulate_steps_kernel msm_accumulate_kernel ntt_step_kernel msm_finalize msm_row msm_top_kernel qap_kernel >> $O/pmc_raw_bls12_381.txt
  rm -rf $O/pmc
done
python tools/pmc_json.py $O/pmc_raw_bls12_381.txt "$tag" bls12_381 > $O/pmc_bls12_381_accumulate.json
# config 5's curve and data path: BLS12-381 2^20 (timed, parity on the timed instance) and the sharded proof over 8 shard
# keys in this process against the oracle
(timeout 300 python bench.py --curve bls12_381 --log-m 20 --steps 5 --warmup 2 --no-extras) > $O/bench_line_bls12_381_2e20.json 2>> $O/bench.err
(timeout 300 python bench.py --curve bls12_381 --log-m 20 --shards-in-process 8 --steps 3) > $O/bench_shards_in_process_bls12_381_2e20.json 2>> $O/bench.err
(timeout 120 python tools/shard_timing.py 20 10 bn254 1,2,4,8 2>&1 | grep -E "^world|per_rank") > $O/shard_timing.txt
(DG16_OVERLAP=1 timeout 120 python tools/shard_timing.py 20 10 bn254 1,2,4,8 2>&1 | grep -E "^world|per_rank") >> $O/shard_timing.txt
(timeout 120 python tools/config4_timing.py 2>&1 | tail -1) > $O/config4_timing.txt
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; uptime) > $O/box.txt 2>&1
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_check'], d['roofline']['kernel_ms'], d['g1_accumulate_ms'],
      d['valu_roofline']['frac'], d['valu_roofline']['whole_proof_valu_frac'])
PY
