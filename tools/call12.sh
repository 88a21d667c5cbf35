#!/bin/bash
# round 6, GPU call 12: GELU' as one byte (SFRON_EPI_GELU_Q / _DGELU_Q): parity, then the step with and without it on one box
set -o pipefail
out=gpurun_out/r06k; mkdir -p $out
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x -s -k "one_byte or fused_epilogues" 2>&1 | grep -E "passed|failed|rel-L2|Error" | tail -6
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fp8.py tests/test_gpu_reference_fixtures.py tests/test_gpu_fisher_and_acceptance.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|worst|Error|assert" | tail -8
python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|DiT-|held-out|Error|assert" | tail -24
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
