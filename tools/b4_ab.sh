for r in 1 2; do for v in product cap2 cap3 cap4 cap6; do
  if [ $v = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  echo -n "$v: "; python3 tools/bench_dit_b4.py 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done; done
