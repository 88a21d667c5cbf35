#!/bin/bash
# rocprofv3 kernel stats of any python tool: tools/prof_cmd.sh <tag> <N rows> <script.py> [args...]   (GPU box only)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; N=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/"$@" > $OUT/out.txt 2> $OUT/err.txt
python3 $ROOT/tools/top_kernels.py "$(find $OUT -name '*kernel_stats.csv' | head -1)" $N
tail -3 $OUT/out.txt | cut -c1-300
