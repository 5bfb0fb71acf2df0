ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r04_inflight
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --no-configs --no-extras --no-cpu-baseline --steps 20 --min-wall 1 > $OUT/trace.log 2>&1
cd $ROOT
python tools/trace_overlap.py $OUT/trace $OUT/overlap.txt > $OUT/overlap.log 2>&1
head -3 $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/trace_head.txt
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_inflight.csv \;
grep -a "^{" $OUT/trace.log | tail -1 | cut -c1-600 > $OUT/bench_line.txt
rm -rf $OUT/trace
cat $OUT/overlap.txt | tail -6; cat $OUT/trace_head.txt | cut -c1-600; cat $OUT/bench_line.txt
