#!/bin/bash
# Wide sort, how the partition bits are split over the two levels: parity tests of the sort (incl. up to 4096
# level-2 bins on few rows), then 2e9 / 2^28 rows end to end with level 2 taking up to 12 / 11 / 10 bits and the even
# split (0), and a kernel trace of the default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_ah}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_b2max=12"
for rows in 2000000000 268435457; do
  ROWS=$rows timeout 400 python scripts/exp_knobs.py sort "" "sort_msd_wide_b2max=11" "sort_msd_wide_b2max=10" "sort_msd_wide_b2max=0" 2>/dev/null | tee -a $OUT/ab.txt
done
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o sort -- python scripts/prof_sort_groupby.py sort 3 > /dev/null 2> $OUT/err.txt
python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) msd > $OUT/sort_kernels.txt 2>&1
head -30 $OUT/sort_kernels.txt
