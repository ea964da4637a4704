#!/bin/bash
# libarrow_amd.so variants that differ in selection.hip's kSelNt (cache policy of the one-pass streams): build/variants/libarrow_amd_selnt<k>.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
for k in "$@"; do
  sed "s/^constexpr int kSelNt = [0-9]*;/constexpr int kSelNt = $k;/" arrow_amd/csrc/selection.hip > arrow_amd/csrc/selection_variant_tmp.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c arrow_amd/csrc/selection_variant_tmp.hip -o build/vobj/selection_$k.o
  rm -f arrow_amd/csrc/selection_variant_tmp.hip
  objs=$(ls build/obj/*.o | grep -v "/selection.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libarrow_amd_selnt$k.so $objs build/vobj/selection_$k.o
  echo "built build/variants/libarrow_amd_selnt$k.so"
done
