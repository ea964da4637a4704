#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_k}
mkdir -p $OUT
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
cp build/variants/libarrow_amd_prof.so arrow_amd/libarrow_amd.so
timeout 300 python scripts/exp_gbp_profile.py 2>&1 | tee $OUT/gbp_tile_phases.txt
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
