#!/bin/bash
# One GPU call: attention forward forms alone, then the step with the round-5 form (4) against the rule (0), alternating.
set -o pipefail
out=gpurun_out/$1; mkdir -p $out
python3 tools/bench_attn.py 2>&1 | tee $out/attn_alone.txt
for rep in 1 2; do
  for v in 0 4; do
    echo "=== SFRON_BENCH_ATTN_FWD=$v (rep $rep)" | tee -a $out/log.txt
    SFRON_BENCH_ATTN_FWD=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>$out/attn_${v}_$rep.err | tee $out/attn_${v}_$rep.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2))
" | tee -a $out/log.txt
  done
done
