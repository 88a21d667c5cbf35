#!/bin/bash
# One GPU call: bench.py for several (library build, extra bench flags) pairs on the SAME box.
#   tools/ab_bench.sh <outdir> "<lib suffix|product>[:flag[,flag...]]" ...
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
for spec in "$@"; do
  v=${spec%%:*}; flags=""
  if [[ "$spec" == *:* ]]; then flags=$(echo "${spec#*:}" | tr ',' ' '); fi
  if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  tag=$(echo "$spec" | tr -c 'A-Za-z0-9_\n' '_')
  echo "=== $spec" | tee -a $out/log.txt
  timeout -k 10 300 python3 tools/bench_ab.py --steps 15 --warmup 4 --no-cpu-baseline $flags 2>$out/bench_$tag.err | tee $out/bench_$tag.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2), 'wgrad us', round(d['roofline']['avg_launch_ms'] * 1e3, 1))
" | tee -a $out/log.txt
done
