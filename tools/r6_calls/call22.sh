# round 6, call 22: timelines of config 4 and of an 8-shard rank at the current library
tag=${1:-r6dd}
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
prof config4 2.0 python tools/config4_timing.py
DG16_OVERLAP=0 prof shard8 3.0 python tools/shard_timing.py 20 4 bn254 8
