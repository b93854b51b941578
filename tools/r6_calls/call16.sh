# round 6, call 16: timelines after the lane reductions -- config 4, an 8-shard rank, small plain MSMs
tag=${1:-r6u}
O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
prof() {  # name window_ms command...
  name=$1; win=$2; shift 2
  rm -rf $O/prof
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- "$@" > $O/${name}_cmd.txt 2>&1
  db=$(find $O/prof -name "*.db" | head -1)
  python tools/rocprof_timeline.py "$db" $win $O/timeline_${name}.md > /dev/null 2>&1
  rm -rf $O/prof
}
prof config4 2.1 python tools/config4_timing.py
DG16_OVERLAP=0 prof shard8 3.2 python tools/shard_timing.py 20 4 bn254 8
prof msm_bn254_g1_2e10 0.45 python tools/msm_small_probe.py bn254 1 10 5
prof msm_bls12_377_g1_2e12 1.0 python tools/msm_small_probe.py bls12_377 1 12 5
ls $O
