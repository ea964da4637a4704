#!/bin/bash
# Sampled level-1 histogram of the wide sort: parity tests, then A/B against the exact histogram at 2e9 and 2^28 rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_m}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_sample_shift=4"
timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_wide_sample_shift=0" "sort_msd_wide_sample_shift=3" "sort_msd_wide_sample_shift=5" > $OUT/ab_2e9.txt 2> $OUT/ab_2e9.err; echo "ab rc=$?"; cat $OUT/ab_2e9.txt
ROWS=268435457 timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_wide_sample_shift=0" > $OUT/ab_2e28.txt 2> $OUT/ab_2e28.err; echo "ab rc=$?"; cat $OUT/ab_2e28.txt
