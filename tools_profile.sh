#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's roofline figures on the GPU box:
#   tools_profile.sh <tag>      -> gpurun_out/prof_<tag>/*  (copy summaries into profiles/)
# Kernel trace + stats in one run; PMC counters each in their own run (never mixed
# with tracing domains other than kernel-trace).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.csv" | head -50
python $ROOT/tools_profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
