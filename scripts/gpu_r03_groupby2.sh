#!/bin/bash
# Round 3: wide group-by after U = 8 / 2^20-row aggregate units; slice-size A/B; scatter micro-benchmark with small LDS chunks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_groupby2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groupby or hash_sum or group_by" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
export DEFAULTS="groupby_wide=1 groupby_wide_agg_chunk_rows=1048576 groupby_max_slice_rows=1073741824 groupby_agg_chunk_rows=262144"
timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" "groupby_max_slice_rows=2147483648" "groupby_max_slice_rows=4278190080" "groupby_max_slice_rows=4278190080 groupby_wide_agg_chunk_rows=2097152" 2> $OUT/knobs_1e7.err | tee $OUT/knobs_1e7.txt
GROUPS=1000000 timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" "groupby_max_slice_rows=4278190080" 2> $OUT/knobs_1e6.err | tee $OUT/knobs_1e6.txt
timeout 200 build/wide_scatter_bench 30 o > $OUT/wide_scatter_small_chunks.txt 2>&1; cat $OUT/wide_scatter_small_chunks.txt
