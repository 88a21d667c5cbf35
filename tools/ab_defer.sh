#!/bin/bash
# One GPU call: the beside-forward sweep launched behind the forward pass's conditioning prologue (round 6) against before it (round 5),
# alternating on one box; then the default run with the legs in the order that showed the third-runner slowdown.
set -o pipefail
out=gpurun_out/$1; mkdir -p $out
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = (d['roofline']['others'].get('configs') or {})
        print('   ms_per_step', round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c.items() if isinstance(v, dict) and 'ms_per_step' in v})
"; }
for rep in 1 2; do
  for v in 1 0; do
    echo "=== SFRON_BENCH_DEFER_SWEEP=$v (rep $rep)" | tee -a $out/log.txt
    SFRON_BENCH_DEFER_SWEEP=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>$out/defer_${v}_$rep.err | tee $out/defer_${v}_$rep.json | line | tee -a $out/log.txt
  done
done
echo "=== default run, legs config2,config5" | tee -a $out/log.txt
SFRON_BENCH_LEGS=config2,config5 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/legs.err | tee $out/legs.json | line | tee -a $out/log.txt
