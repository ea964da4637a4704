#!/bin/bash
# Round 6 call A: does a flat level whose every store is a whole 128-byte line reach the linear write rate?
# scripts/micro/wc_lines_bench.hip (the dense-range lines plan: write-combined scatter + direct-indexed aggregate) at 2^30
# and 4e9 rows, and the hash_sum leg of the bench on the same box as the reference.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_a}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 ./build/wc_lines_bench 30 10000000 12288 > $OUT/wc_lines_2e30_w12288.txt 2>&1; echo "rc=$?"; cat $OUT/wc_lines_2e30_w12288.txt
timeout 300 ./build/wc_lines_bench 30 9000000 8192 > $OUT/wc_lines_2e30_w8192.txt 2>&1; echo "rc=$?"; cat $OUT/wc_lines_2e30_w8192.txt
timeout 400 ./build/wc_lines_bench 32 10000000 12288 4000000000 > $OUT/wc_lines_4e9_w12288.txt 2>&1; echo "rc=$?"; cat $OUT/wc_lines_4e9_w12288.txt
timeout 600 python bench.py --workload hash_sum --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_hash_sum.json 2> $OUT/bench_hash_sum.err; echo "bench rc=$?"; cat $OUT/bench_hash_sum.json | head -c 1500
