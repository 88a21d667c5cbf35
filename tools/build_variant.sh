#!/bin/bash
# Same-box A-B builds: tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
# compiles the named sources (default: gemm.hip) with the extra flags into csrc/var_<name>/ and links them with the product
# objects of the other sources into unified-unlearning-w-remain-geometry_amd/libsfron_<name>.so (git-ignored; ships with gpurun).
# Tools load it through SFRON_LIB_NAME=libsfron_<name>.so (tools/ab_lib.py); the product never does.
set -e
name=$1; flags=$2; shift 2 || true
files=${@:-gemm.hip}
cd "$(dirname "$0")/../unified-unlearning-w-remain-geometry_amd/csrc"
make -j8 >/dev/null
mkdir -p var_$name
objs=""
for f in *.hip; do
  o=${f%.hip}.o
  if [[ " $files " == *" $f "* ]]; then
    # the Makefile's CXXFLAGS, per-file additions included (attn.hip is built with the VGPR form of the MFMA results: an A-B of an
    # attention change must not also switch that -- ADVICE r4)
    extra=""
    if [ "$f" = attn.hip ]; then extra="-mllvm -amdgpu-mfma-vgpr-form=1"; fi
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable $extra $flags -c $f -o var_$name/$o &
    objs="$objs var_$name/$o"
  else
    objs="$objs $o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfron_$name.so $objs
echo built ../libsfron_$name.so
