#!/bin/bash
# One GPU call: bench.py of THIS tree under several values of one bench knob (environment variable), alternating on one box.
#   tools/ab_env.sh <outdir> <VAR> <value> [<value> ...]
out=gpurun_out/$1; var=$2; shift 2
mkdir -p $out
for v in "$@" "$@"; do
  echo "=== $var=$v" | tee -a $out/log.txt
  env $var=$v timeout -k 10 300 python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-configs 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2))
" | tee -a $out/log.txt
done
