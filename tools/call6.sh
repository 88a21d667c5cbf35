#!/bin/bash
# round 6, GPU call 6: hardware-queue count (GPU_MAX_HW_QUEUES 4 = default vs 8): headline A-B and the DP footprint table under both
set -o pipefail
out=gpurun_out/r06f; mkdir -p $out
python -m pytest tests/test_gpu_reference_fixtures.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
  for q in 4 8; do
    echo "=== GPU_MAX_HW_QUEUES=$q (rep $rep)" | tee -a $out/log.txt
    GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>$out/q${q}_$rep.err | tee $out/q${q}_$rep.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step', round(d['ms_per_step'], 2))
" | tee -a $out/log.txt
  done
done
for q in 4 8; do
  echo "=== DP footprint, GPU_MAX_HW_QUEUES=$q" | tee -a $out/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python3 tools/bench_dp_footprint.py 10 > $out/dp_footprint_q$q.txt 2>&1; tail -10 $out/dp_footprint_q$q.txt | tee -a $out/log.txt
done
