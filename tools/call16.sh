#!/bin/bash
# round 6, GPU call 16: timing experiment -- the saved branch outputs of proj / fc2 (bf16 [M x D], read by the gate backward only): what do they cost?
set -o pipefail
out=gpurun_out/r06m; mkdir -p $out
export SFRON_BENCH_ZERO_WS=1
for rep in 1 2; do
  for v in product nobr; do
    if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
    echo "=== $v (rep $rep)"
    timeout -k 10 300 python3 tools/bench_ab.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2), ' finite', d['finite_losses'])
"
  done
done
