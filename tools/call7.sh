#!/bin/bash
# round 6, GPU call 7: probed streams + caller-owned fp8 range words: DiT / DP / fp8 tests, headline, DP footprint table
set -o pipefail
out=gpurun_out/r06g; mkdir -p $out
python -m pytest tests/test_gpu_dit.py tests/test_gpu_dp_overlap.py tests/test_gpu_fp8.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>$out/bench_$rep.err | tee $out/bench_$rep.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   headline ms_per_step', round(d['ms_per_step'], 2))
"
done
timeout -k 10 400 python3 tools/bench_dp_footprint.py 10 > $out/dp_footprint.txt 2>&1; tail -10 $out/dp_footprint.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench_full.err | tee $out/bench_full.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['roofline']['others']['configs']
        print('   full run: headline', round(d['ms_per_step'], 2), {k: (round(v['ms_per_step'], 2) if 'ms_per_step' in v else {kk: round(vv['ms_per_iteration'], 1) for kk, vv in v.items() if isinstance(vv, dict)}) for k, v in c.items() if isinstance(v, dict)})
"
