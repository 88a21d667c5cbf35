#!/bin/bash
# In-step per-kernel averages of the headline bench (rocprofv3 kernel trace): tools/step_kernels.sh [tag] [N rows] [bench args...]
# writes gpurun_out/<tag>/..._kernel_stats.csv and prints the top N rows.  GPU box only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-stepk}; N=${2:-16}; shift 2 2>/dev/null
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 $ROOT/tools/top_kernels.py "$(find $OUT -name '*kernel_stats.csv' | head -1)" $N
cut -c1-230 $OUT/bench.json
