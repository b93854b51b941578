# SQ counter evidence for the bucket accumulations (one gpurun call):  bash tools/sq_evidence.sh <tag>
#   gpurun_out/<tag>/pmc_sq_bn254.md, pmc_sq_bls12_381.md   (tools/pmc_sq_report.py over separate --pmc passes)
# Counters in their own runs with --kernel-trace only (no --stats / --sys-trace next to --pmc).
tag=${1:-sq}; O=gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM"
P3="SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
for curve in bn254 bls12_381; do
  dirs=""
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1)); d=$O/pmc_${curve}_$i; rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $d -o run -- python tools/shard_timing.py 20 2 $curve 1 > $O/pmc_${curve}_$i.log 2>&1
    echo "pass $curve $i rc $?" >> $O/passes.txt
    dirs="$dirs $d"
  done
  python tools/pmc_sq_report.py $O/pmc_sq_$curve.md "2^20-constraint $curve proof x 5 (rocprofv3 --kernel-trace --pmc <pass> --output-format csv -- python tools/shard_timing.py 20 2 $curve 1; passes: [$P1] [$P2] [$P3])" $dirs > /dev/null 2>> $O/passes.txt
  for d in $dirs; do rm -rf $d; done
done
