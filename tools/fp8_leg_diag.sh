#!/bin/bash
# One GPU call (VERDICT r5 #2i): why does the fp8 leg of bench.py read slower than `bench.py --fp8` alone?
#   tools/fp8_leg_diag.sh <outdir>
# (a) the headline alone at 20 / 100 / 300 timed steps (does the number drift with the length of the run?), (b) --fp8 alone at 20 / 100,
# (c) the default run with the fp8 leg FIRST / LAST among the legs.
set -o pipefail
out=gpurun_out/$1; mkdir -p $out
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = (d['roofline']['others'].get('configs') or {})
        print('   ms_per_step', round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c.items() if isinstance(v, dict) and 'ms_per_step' in v})
"; }
for n in 20 100 300; do
  echo "=== bf16 alone, $n timed steps" | tee -a $out/log.txt
  timeout -k 10 300 python3 bench.py --steps $n --warmup 6 --no-configs --no-cpu-baseline 2>$out/bf16_$n.err | tee $out/bf16_$n.json | line | tee -a $out/log.txt
done
for n in 20 100; do
  echo "=== fp8 alone, $n timed steps" | tee -a $out/log.txt
  timeout -k 10 300 python3 bench.py --fp8 --steps $n --warmup 6 --no-configs --no-cpu-baseline 2>$out/fp8_$n.err | tee $out/fp8_$n.json | line | tee -a $out/log.txt
done
for order in config5,config2 config2,config5; do
  echo "=== default run, legs $order" | tee -a $out/log.txt
  SFRON_BENCH_LEGS=$order timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/legs_$order.err | tee $out/legs_$order.json | line | tee -a $out/log.txt
done
