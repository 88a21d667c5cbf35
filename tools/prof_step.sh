#!/bin/bash
# One GPU call: kernel trace of the headline step -> per-class table, two-stream timeline, the two stage boundaries (tools/trace_boundary.py).
#   tools/prof_step.sh <tag> [extra bench flags]      -> gpurun_out/<tag>/<tag>_{bench_kernel_stats.csv,kernel_table.md,timeline.txt,stage_boundary.txt}
set -e
TAG=$1; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs "$@" > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
S=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); T=$(ls $OUT/stats/*/*kernel_trace.csv | head -1)
cp $S $OUT/${TAG}_bench_kernel_stats.csv
python3 $ROOT/tools/kernel_table.py $OUT/${TAG}_bench_kernel_stats.csv --md $OUT/${TAG}_kernel_table.md --json $OUT/${TAG}_kernel_table.json > /dev/null
python3 $ROOT/tools/trace_timeline.py $T 60 > $OUT/${TAG}_timeline.txt 2>&1 || true
python3 $ROOT/tools/trace_boundary.py $T > $OUT/${TAG}_stage_boundary.txt 2>&1 || true
rm -rf $OUT/stats
