#!/bin/bash
# One GPU call: GEMM micro-benchmarks and bench.py for several builds of the library on the SAME box.
#   tools/ab_run.sh <outdir> <lib suffix> [<lib suffix> ...]     ("product" = libsfron.so)
set -e -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  echo "=== $v: bench_gemm" | tee -a $out/log.txt
  timeout -k 10 300 python3 tools/bench_gemm.py 2>&1 | tee $out/gemm_$v.txt
done
for v in "$@"; do
  if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  echo "=== $v: bench.py" | tee -a $out/log.txt
  timeout -k 10 300 python3 tools/bench_ab.py --steps 15 --warmup 4 --no-cpu-baseline 2>$out/bench_$v.err | tee $out/bench_$v.json
done
