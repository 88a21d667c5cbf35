#!/bin/bash
# bench.py of the baseline worktree (_ab/base: last round's tree with its own library): prints ms_per_step
out=gpurun_out/$1; mkdir -p $out
( cd _ab/base && timeout -k 10 300 python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline $2 2>../../$out/bench_old.err ) | tee $out/bench_old.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('=== OLD tree   ms_per_step', round(d['ms_per_step'], 2), 'wgrad us', round(d['roofline']['avg_launch_ms'] * 1e3, 1))
" | tee -a $out/log.txt
