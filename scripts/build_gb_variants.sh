#!/bin/bash
# libarrow_amd.so variants that differ in groupby.hip's kGbNt (cache policy of the wide form's streams): build/variants/libarrow_amd_gbnt<k>.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
for k in "$@"; do
  sed "s/^constexpr int kGbNt = [0-9]*;/constexpr int kGbNt = $k;/" arrow_amd/csrc/groupby.hip > arrow_amd/csrc/groupby_variant_tmp.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c arrow_amd/csrc/groupby_variant_tmp.hip -o build/vobj/groupby_$k.o
  rm -f arrow_amd/csrc/groupby_variant_tmp.hip
  objs=$(ls build/obj/*.o | grep -v "/groupby.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libarrow_amd_gbnt$k.so $objs build/vobj/groupby_$k.o
  echo "built build/variants/libarrow_amd_gbnt$k.so"
done
