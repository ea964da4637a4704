#!/bin/bash
# Round 5 call E: the give-up check of the rec8 finish is workgroup-uniform now: the failing sequence ten times over,
# then the rec8 + wide sort tests and the full-size configs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_e}
mkdir -p $OUT
ulimit -c 0
BASE="sort_msd=1 sort_msd_segment_rows=4096 sort_msd_wide_bits=18 sort_msd_wide_b2max=9 sort_msd_wide_gap2=0 sort_msd_wide_sample_shift=0"
for o in "sort_msd_wide_rec8_tie_shift=40" "sort_msd_wide_rec8_tie_shift=40 sort_msd_wide_wc=0" "sort_msd_wide_rec8_tie_shift=40 sort_msd_wide_rpt1=16 sort_msd_wide_wc_prefetch=0"; do
  echo "== $o"
  timeout 200 python scripts/exp_sort_rec8_step.py 20000003 "$BASE $o" 10 2>&1 | grep -v "^  File\|Extension modules\|^$\|amdgpu.ids" | tail -2
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -x --durations=5 -k "sort_wide or config5 or sort_msd" > $OUT/pytest_sort.log 2>&1; echo "pytest sort rc=$?"; tail -9 $OUT/pytest_sort.log
