mkdir -p gpurun_out
ROOT=$(pwd)
for cfg in 1 2 3 4; do bash tools/profile.sh r04c$cfg $cfg > gpurun_out/prof_r04c$cfg.log 2>&1; done
# the mode that produces `value` -- 12 independent jobs in flight on 12 streams
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r04_inflight
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --no-configs --no-extras --no-cpu-baseline --steps 20 --min-wall 1 > $OUT/trace.log 2>&1
cd $ROOT
python tools/trace_overlap.py $OUT/trace $OUT/overlap.txt > $OUT/overlap.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_inflight.csv \;
tail -2 $OUT/trace.log > $OUT/bench_line.txt
rm -rf $OUT/trace
cat $OUT/overlap.txt
for cfg in 1 2 3 4; do tail -5 gpurun_out/prof_r04c$cfg.log; done
du -sh gpurun_out
