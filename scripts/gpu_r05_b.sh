#!/bin/bash
# Round 5 call B: rec8 with the per-bucket tie budget + the write-combined level 1: tests, kernel trace, A/B of the forms.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=5 -k "sort_wide or config5" > $OUT/pytest_sort.log 2>&1; echo "pytest sort rc=$?"; tail -9 $OUT/pytest_sort.log
RUN_TAG=${RUN_TAG:-r05_b}/sg WHAT=sort bash scripts/gpu_prof_sg.sh
for o in "sort_msd_wide_rpt1=16" "sort_msd_wide_wc=0" "sort_msd_wide_rec8=0" "sort_msd_wide_wc=128" "sort_msd_wide_b2max=10" "sort_msd_wide_b2max=10 sort_msd_wide_rpt1=16" "sort_msd_bucket_cpt=8" "sort_msd_tiny_bucket=1"; do
  echo "== $o"
  ARX_OPTIONS="$o" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run [12]"
done
