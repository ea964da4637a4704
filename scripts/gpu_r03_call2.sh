#!/bin/bash
# Round 3, GPU call 2: the flat 2048-bin level again, with register-staged big tiles (longer runs) and non-temporal input
# loads (wide_scatter_bench "big" mode).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_call2
mkdir -p $OUT
for lg in 28 30; do
  echo "== wide_scatter_bench big 2^$lg"; timeout 300 build/wide_scatter_bench $lg big > $OUT/wide_scatter_big_$lg.txt 2>&1; echo "rc=$?"; cat $OUT/wide_scatter_big_$lg.txt
done
