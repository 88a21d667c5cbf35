#!/bin/bash
# One call on the GPU box: the rocprofv3 summaries that DESIGN.md section 6 and bench.py's roofline quote, under gpurun_out/<tag>/.
#   bash tools/collect_profiles.sh r03      (then copy what is to be judged into profiles/)
set -e
TAG=${1:-rNN}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# (1) kernel stats + trace of the headline command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
S=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); T=$(ls $OUT/stats/*/*kernel_trace.csv | head -1)
cp $S $OUT/${TAG}_bench_kernel_stats.csv
python3 $ROOT/tools/kernel_table.py $OUT/${TAG}_bench_kernel_stats.csv --md $OUT/${TAG}_kernel_table.md --json $OUT/${TAG}_kernel_table.json > /dev/null
python3 $ROOT/tools/trace_timeline.py $T 60 > $OUT/${TAG}_timeline.txt 2>&1 || true
rm -rf $OUT/stats
# (2) PMC traffic of the probed kernels (FETCH_SIZE / WRITE_SIZE in separate passes)
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc $TAG > $OUT/pmc_traffic.log 2>&1 || tail -5 $OUT/pmc_traffic.log
cp $OUT/pmc/${TAG}_*traffic*.json $OUT/pmc/${TAG}_pmc_hbm_traffic_per_kernel.csv $OUT/ 2>/dev/null || true
rm -rf $OUT/pmc
# the bench lines below report `traffic` only from profiles/*_traffic.json of the SAME sources: put this run's there (on this box's copy)
cp $OUT/${TAG}_*traffic*.json $ROOT/profiles/ 2>/dev/null || true
# (2b) kernel stats of the fp8 run (config 5)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats8 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --fp8 > /dev/null 2>&1 || true
S8=$(ls $OUT/stats8/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S8" ] && cp $S8 $OUT/${TAG}_fp8_bench_kernel_stats.csv; rm -rf $OUT/stats8
python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --fp8 > $OUT/${TAG}_bench_fp8.json 2> /dev/null || true
# (3) the plain bench line (no profiler)
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
ls $OUT
