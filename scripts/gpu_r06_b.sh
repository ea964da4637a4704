#!/bin/bash
# Round 6 call C: wc_scatter2 (bank-conflict-free LDS layout, prefetched chunk reservations, carried rows)
# atomics-bound — single-lane returning atomics queue up per LINE at ~110 M/s) and chunks of 4 / 8 / 16 lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_c}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 ./build/wc_lines_bench 30 10000000 12288 > $OUT/wc_lines_2e30_w12288.txt 2>&1; echo "rc=$?"; cat $OUT/wc_lines_2e30_w12288.txt
timeout 400 ./build/wc_lines_bench 32 10000000 12288 4000000000 > $OUT/wc_lines_4e9_w12288.txt 2>&1; echo "rc=$?"; cat $OUT/wc_lines_4e9_w12288.txt
