#!/bin/bash
# round 6, GPU call 10: timing experiment -- fc1 + GELU without its second (pre-activation) output: what is the double write worth in the step?
set -o pipefail
out=gpurun_out/r06j; mkdir -p $out
export SFRON_BENCH_ZERO_WS=1
for rep in 1 2; do
  for v in product nohpre; do
    if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
    echo "=== $v (rep $rep)"
    timeout -k 10 300 python3 tools/bench_ab.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); o = d['roofline']['others']['fwd_fc1_gelu']
        print('   ms_per_step', round(d['ms_per_step'], 2), ' fc1+GELU block 0 us', round(o['avg_launch_ms'] * 1e3, 1), ' finite', d['finite_losses'])
"
  done
done
