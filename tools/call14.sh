#!/bin/bash
set -o pipefail
out=gpurun_out/r06l; mkdir -p $out
python3 tools/diag_geluq.py 2>&1 | tail -7
echo "=== soak 12 x 2, product"; python3 tools/soak_repro.py 12 2>&1 | tail -1
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x -s -k "one_byte" 2>&1 | grep -E "passed|failed|rel-L2|Error" | tail -5
python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x -k "bit_for_bit or gemms" 2>&1 | tail -2
for rep in 1 2; do
  for v in product nogq; do
    if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
    echo "=== $v (rep $rep)"
    timeout -k 10 300 python3 tools/bench_ab.py --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); o = d['roofline']['others']['fwd_fc1_gelu']
        print('   ms_per_step', round(d['ms_per_step'], 2), ' fc1+GELU block 0 us', round(o['avg_launch_ms'] * 1e3, 1), ' finite', d['finite_losses'])
"
    timeout -k 10 300 python3 tools/bench_ab.py --fp8 --steps 20 --warmup 6 --no-configs --no-cpu-baseline 2>>$out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   fp8 ms_per_step', round(d['ms_per_step'], 2))
"
  done
done
