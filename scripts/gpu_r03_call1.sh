#!/bin/bash
# Round 3, GPU call 1: three micro-benchmarks that decide this round's kernel work —
#  (a) l2_atomic_bench: the group-by's "global atomics = 12 Grows/s" assumption, re-measured (non-returning, XCD-affine slices)
#  (b) wide_scatter_bench: can ONE flat 2048/8192-bin level with per-XCD frontiers replace the two partition levels
#  (c) stream_bench: the box's copy ceiling and the cast / greater forms against it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_call1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== l2_atomic_bench"; timeout 200 build/l2_atomic_bench 28 > $OUT/l2_atomic_bench.txt 2>&1; echo "rc=$?"; cat $OUT/l2_atomic_bench.txt
echo "== wide_scatter_bench 2^28"; timeout 200 build/wide_scatter_bench 28 > $OUT/wide_scatter_bench_28.txt 2>&1; echo "rc=$?"; cat $OUT/wide_scatter_bench_28.txt
echo "== wide_scatter_bench 2^30"; timeout 200 build/wide_scatter_bench 30 > $OUT/wide_scatter_bench_30.txt 2>&1; echo "rc=$?"; cat $OUT/wide_scatter_bench_30.txt
echo "== stream_bench"; timeout 200 build/stream_bench 30 > $OUT/stream_bench.txt 2>&1; echo "rc=$?"; grep -E "one-shot|grid=     2048 |Memcpy" $OUT/stream_bench.txt | sort -k9 -n -r | head -60
