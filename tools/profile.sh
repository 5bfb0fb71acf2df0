#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's roofline figures on the GPU box:
#   tools/profile.sh <tag> [config]   -> gpurun_out/prof_<tag>/*   (tools/profile_install.py copies the
#                                        summaries into profiles/<round>/ and merges pmc_traffic.json)
# config = 1 (default: BASELINE configs[1], one 48 kHz x 10 s job, one stream), 2, 3 or 4 (bench.py --only-config).
# Kernel trace + stats in one run; PMC counters each in their own run (never mixed with tracing domains
# other than kernel-trace: MI355X_MICROARCH.md / gpurun's rule).
set -u
TAG=${1:-r02}
CFG=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "1" ]; then
  CMD="python $ROOT/bench.py --steps 5 --warmup 2 --streams 1 --min-wall 0 --no-cpu-baseline --no-extras --no-configs"
  FRAMES=2001
else
  CMD="python $ROOT/bench.py --only-config $CFG --steps 2 --min-wall 0 --contexts ${CTX:-1}"
  case $CFG in 2) FRAMES=256256;; 3) FRAMES=128128;; 4) FRAMES=64064;; esac
fi
echo "$CMD" > $OUT/command.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
tail -3 $OUT/trace.log
PROFILE_FRAMES=$FRAMES PROFILE_CONFIG=$CFG python $ROOT/tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
# keep the evidence small: the stats CSV and per-kernel averages of every PMC pass (a batch run has tens of
# thousands of dispatch rows; gpurun merges at most 64 MiB back)
mkdir -p $OUT/keep
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/keep/ \;
python $ROOT/tools/profile_aggregate.py $OUT
rm -rf $OUT/trace $OUT/pmc_*/ $OUT/*.log
du -sh $OUT
