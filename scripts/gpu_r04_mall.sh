#!/bin/bash
# Round 4: does a chunk-sized produce -> consume pair run faster than whole passes (memory-side cache residency)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_v
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 build/mall_bench ${LG:-33} > $OUT/mall_bench.txt 2>&1; echo "rc=$?"; cat $OUT/mall_bench.txt
