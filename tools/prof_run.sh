# usage: bash tools/prof_run.sh <tag> <env assignments or ""> <command...>   -> gpurun_out/<tag>_stats.md (+ timeline of the last proof)
tag=$1; shift; envs=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
out=gpurun_out/prof_$tag
rm -rf $out
env $envs rocprofv3 --kernel-trace --stats -d $out -o run -- "$@" > gpurun_out/${tag}_cmd.txt 2>&1
db=$(find $out -name "*.db" | head -1)
python tools/rocprof_stats.py "$db" gpurun_out/${tag}_stats.md > /dev/null
python tools/rocprof_timeline.py "$db" 45 gpurun_out/${tag}_timeline.md > /dev/null 2>&1
rm -rf $out
