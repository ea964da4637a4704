#!/bin/bash
# Round 3: microbenchmark of a sweeping filter without LDS (coalesced 16-byte loads, ballot ranks, direct stores).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_w}
mkdir -p $OUT
timeout 200 build/sweep_compact_bench 30 | tee $OUT/sweep_compact_bench.txt
