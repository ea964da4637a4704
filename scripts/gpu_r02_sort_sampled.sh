#!/bin/bash
# Wide sort with estimated bucket sizes (sampled level-1 histogram, fixed level-2 rooms) and 12-byte records:
# parity tests, then A/B against the exact forms at 2e9 and 2^28 rows, then a kernel trace of the 2e9 sort.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_m}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_sample_shift=4 sort_msd_wide_gap2=1"
timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_wide_gap2=0" "sort_msd_wide_sample_shift=0 sort_msd_wide_gap2=0" > $OUT/ab_2e9.txt 2> $OUT/ab_2e9.err; echo "ab rc=$?"; cat $OUT/ab_2e9.txt
ROWS=268435457 timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_wide_gap2=0" "sort_msd_wide_sample_shift=0 sort_msd_wide_gap2=0" > $OUT/ab_2e28.txt 2> $OUT/ab_2e28.err; echo "ab rc=$?"; cat $OUT/ab_2e28.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python scripts/exp_knobs.py sort "" > /dev/null 2> $OUT/prof.err; echo "prof rc=$?"
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -14 $OUT/kernel_stats.txt
find $OUT -name "*.db" -delete
