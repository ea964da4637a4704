#!/bin/bash
# Round 3: the wide one-level group-by form on the GPU — parity tests, end-to-end A/B against the two-level plan at
# 4e9 rows (1e7 and 1e6 keys), its knobs, and a kernel trace + FETCH/WRITE counters of the default configuration.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_groupby
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groupby or hash_sum or buffer_copy or group_by" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
export DEFAULTS="groupby_wide=1 groupby_wide_agg_chunk_rows=524288 groupby_max_slice_rows=1073741824 groupby_wide_max_bits=11"
timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" "groupby_wide_agg_chunk_rows=262144" "groupby_wide_agg_chunk_rows=1048576" "groupby_wide_agg_chunk_rows=4194304" "groupby_max_slice_rows=2147483648" "groupby_max_slice_rows=4278190080" 2> $OUT/knobs_1e7.err | tee $OUT/knobs_1e7.txt
GROUPS=1000000 timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" 2> $OUT/knobs_1e6.err | tee $OUT/knobs_1e6.txt
GROUPS=100000 ROWS=1073741824 timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" 2> $OUT/knobs_1e5.err | tee $OUT/knobs_1e5.txt
RUN_TAG=r03_groupby/prof PMC=1 WHAT=groupby bash scripts/gpu_prof_sg.sh
