#!/bin/bash
# libarrow_amd.so with groupby.hip's kGbPairAtomics = 0 (one cursor atomic per bin): build/variants/libarrow_amd_gbpair0.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
sed "s/^constexpr int kGbPairAtomics = [0-9]*;/constexpr int kGbPairAtomics = 0;/" arrow_amd/csrc/groupby.hip > arrow_amd/csrc/groupby_variant_tmp.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c arrow_amd/csrc/groupby_variant_tmp.hip -o build/vobj/groupby_pair0.o
rm -f arrow_amd/csrc/groupby_variant_tmp.hip
objs=$(ls build/obj/*.o | grep -v "/groupby.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libarrow_amd_gbpair0.so $objs build/vobj/groupby_pair0.o
echo "built build/variants/libarrow_amd_gbpair0.so"
