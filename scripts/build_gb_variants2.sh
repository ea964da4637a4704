#!/bin/bash
# libarrow_amd.so variants that differ in constexpr switches of groupby.hip: each argument is "name:sed-expression"
# (e.g. "u12:s/kGbWideAggU = 8/kGbWideAggU = 12/") -> build/variants/libarrow_amd_<name>.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
for spec in "$@"; do
  name=${spec%%:*}; expr=${spec#*:}
  sed -e "$expr" arrow_amd/csrc/groupby.hip > arrow_amd/csrc/groupby_variant_tmp_$name.hip
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function ${VARIANT_FLAGS:-} -c arrow_amd/csrc/groupby_variant_tmp_$name.hip -o build/vobj/groupby_$name.o
    rm -f arrow_amd/csrc/groupby_variant_tmp_$name.hip
    objs=$(ls build/obj/*.o | grep -v "/groupby.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libarrow_amd_$name.so $objs build/vobj/groupby_$name.o
    echo "built build/variants/libarrow_amd_$name.so" ) &
done
wait
