#!/bin/bash
# Round 3: the wide group-by without its histogram pass (rooms) — parity tests, A/B against the counted wide plan and the
# two-level plan at 4e9 rows, skewed keys (the overflow fallback), kernel trace + PMC.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_groupby3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groupby or hash_sum or group_by or hash_minmax" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
export DEFAULTS="groupby_wide=1 groupby_wide_rooms=1 groupby_wide_agg_chunk_rows=2097152"
timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide_rooms=0" "groupby_wide=0" "groupby_wide_agg_chunk_rows=4194304" 2> $OUT/knobs_1e7.err | tee $OUT/knobs_1e7.txt
GROUPS=1000000 timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide_rooms=0" "groupby_wide=0" 2> $OUT/knobs_1e6.err | tee $OUT/knobs_1e6.txt
GROUPS=100000 timeout 600 python scripts/exp_knobs.py groupby "" "groupby_wide=0" 2> $OUT/knobs_1e5.err | tee $OUT/knobs_1e5.txt
RUN_TAG=r03_groupby3/prof PMC=1 WHAT=groupby bash scripts/gpu_prof_sg.sh
