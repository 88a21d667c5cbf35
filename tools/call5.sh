#!/bin/bash
# round 6, GPU call 5: gemm parity, gated-residual epilogue A-B, DP footprint table, config-2 queue picture
set -o pipefail
mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x 2>&1 | tail -2
python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x -k "gemms or batch32" 2>&1 | tail -2
python -m pytest tests/test_gpu_reference_fixtures.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|SD |worst" | tail -8
bash tools/ab_bench.sh r06e "product:--no-configs" "gr5:--no-configs" "product:--no-configs" "gr5:--no-configs" 2>&1 | grep -E "===|ms_per_step"
python3 tools/bench_dp_footprint.py 12 > gpurun_out/r06e/dp_footprint.txt 2>&1; tail -12 gpurun_out/r06e/dp_footprint.txt
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r06e/b4 -- python3 $ROOT/tools/bench_dit_b4.py > $ROOT/gpurun_out/r06e/b4_out.txt 2>&1
T=$(ls $ROOT/gpurun_out/r06e/b4/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/trace_queues.py $T k_ema 6 > $ROOT/gpurun_out/r06e/b4_queues.txt 2>&1; cat $ROOT/gpurun_out/r06e/b4_queues.txt | head -50
rm -rf $ROOT/gpurun_out/r06e/b4
