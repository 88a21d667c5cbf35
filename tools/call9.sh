#!/bin/bash
# round 6, GPU call 9: priority of the beside-forward sweep stream (A-B), hardware queue counts 3 / 5 / 6
set -o pipefail
out=gpurun_out/r06i; mkdir -p $out
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { "$@" timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2))
"; }
for rep in 1 2; do
  echo "=== default (rep $rep)"; run env
  echo "=== sweep stream priority 1 / lowest (rep $rep)"; run env SFRON_BENCH_SWEEP_PRIORITY=1
  echo "=== sweep stream priority -1 / highest (rep $rep)"; run env SFRON_BENCH_SWEEP_PRIORITY=-1
done
for q in 3 5 6; do echo "=== GPU_MAX_HW_QUEUES=$q"; run env GPU_MAX_HW_QUEUES=$q; done
