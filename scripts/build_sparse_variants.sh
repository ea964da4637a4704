#!/bin/bash
# libarrow_amd.so variants of the gather-form filter kernel: build/variants/libarrow_amd_sparse_u<U>_w<WINDOW>.so
# usage: build_sparse_variants.sh U:WINDOW ...
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
for uw in "$@"; do
  u=${uw%%:*}; w=${uw##*:}
  sed -e "s/^constexpr int kSparseU = [0-9]*;/constexpr int kSparseU = $u;/" -e "s/^constexpr int kSparseWindow = [0-9]*;/constexpr int kSparseWindow = $w;/" arrow_amd/csrc/selection.hip > arrow_amd/csrc/selection_variant_tmp.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c arrow_amd/csrc/selection_variant_tmp.hip -o build/vobj/selection_u${u}_w${w}.o
  rm -f arrow_amd/csrc/selection_variant_tmp.hip
  objs=$(ls build/obj/*.o | grep -v "/selection.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libarrow_amd_sparse_u${u}_w${w}.so $objs build/vobj/selection_u${u}_w${w}.o
  echo "built build/variants/libarrow_amd_sparse_u${u}_w${w}.so"
done
