#!/bin/bash
# One GPU call: BASELINE config 2 (tools/bench_dit_b4.py) for several library builds, each twice, alternating.  tools/b4_ab.sh product <variant> ...
for r in 1 2; do for v in "$@"; do
  if [ $v = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  echo -n "$v: "; python3 tools/bench_dit_b4.py 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'], 3))"
done; done
