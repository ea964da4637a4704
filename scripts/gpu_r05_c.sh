#!/bin/bash
# Round 5 call C: the write-combined level 1 with prefetch: tests, kernel trace, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_c}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=5 -k "sort_wide_rec8 or config5" > $OUT/pytest_sort.log 2>&1; echo "pytest sort rc=$?"; tail -9 $OUT/pytest_sort.log
RUN_TAG=${RUN_TAG:-r05_c}/sg WHAT=sort bash scripts/gpu_prof_sg.sh
for o in "sort_msd_wide_rpt1=8" "sort_msd_wide_wc_prefetch=0" "sort_msd_wide_wc_prefetch=0 sort_msd_wide_rpt1=16" "sort_msd_wide_wc=512" "sort_msd_wide_wc=248"; do
  echo "== $o"
  ARX_OPTIONS="$o" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run [12]"
done
