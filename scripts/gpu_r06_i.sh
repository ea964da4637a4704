#!/bin/bash
# Round 6 call I: rows per batch of the sort's append kernel (4 / 6 / 8), the wide-form parity grid again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_i}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=4 -k "sort_wide_rec8" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
for r in 4 6 8; do
  ARX_OPTIONS="sort_msd_wide_wc_rows=$r" timeout 300 python scripts/prof_sort_groupby.py sort 3 > $OUT/sort_rows_$r.txt 2>&1; echo "rows=$r rc=$?"; grep "run" $OUT/sort_rows_$r.txt
done
