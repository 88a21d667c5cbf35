#!/bin/bash
# round 6, final collection on the final sources: headline profiles (stats, table, timeline, PMC traffic, fp8), stage boundary, per-config kernel tables
set -o pipefail
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1; tail -3 gpurun_out/r06_collect.log
bash tools/prof_step.sh r06s > /dev/null 2>&1; head -3 gpurun_out/r06s/r06s_stage_boundary.txt
for spec in "r06c1 tools/bench_ddpm.py" "r06c2 tools/bench_dit_b4.py" "r06c4f tools/bench_sd.py" "r06c4x tools/bench_sd.py --method xattn" "r06c4f8 tools/bench_sd.py --batch 8"; do
  set -- $spec; tag=$1; shift
  bash tools/prof_cmd.sh $tag 14 "$@" > gpurun_out/${tag}_kernels.txt 2>&1; tail -2 gpurun_out/${tag}_kernels.txt | cut -c1-200
  rm -rf gpurun_out/$tag
done
