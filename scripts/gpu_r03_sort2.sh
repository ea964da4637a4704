#!/bin/bash
# Round 3, sort: the persistent bucket finish (prefetches its next bucket) against one workgroup per bucket, and how the
# partition bits are split now that level-1 tiles are 24 rows per thread.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_i}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register_staged or sort_wide_sampled or sort_keys_with" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_rpt1=24 sort_msd_wide_rpt2=8 sort_msd_wide_bits=0 sort_msd_tiny_bucket=1 sort_msd_wide_b2max=10 sort_msd_bucket_persist=4"
timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_bucket_persist=0" "sort_msd_bucket_persist=2" "sort_msd_bucket_persist=8" "sort_msd_bucket_persist=16" \
  "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16" "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16 sort_msd_bucket_persist=0" \
  "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16 sort_msd_bucket_persist=8" \
  "sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16" "sort_msd_wide_bits=20 sort_msd_wide_b2max=12 sort_msd_wide_rpt2=16" \
  "sort_msd_wide_bits=20 sort_msd_wide_b2max=11" 2> $OUT/ab_err.txt | tee $OUT/ab.txt
tail -2 $OUT/ab_err.txt
for cfg in "" "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16"; do
  tag=$(echo "${cfg:-defaults}" | tr ' =' '__')
  rm -rf /tmp/prof
  ARX_OPTIONS="$cfg" timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o sort -- python scripts/prof_sort_groupby.py sort 3 > $OUT/run_$tag.txt 2> $OUT/err_$tag.txt
  python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) msd > $OUT/sort_kernels_$tag.txt 2>&1
  echo "== $tag"; head -6 $OUT/sort_kernels_$tag.txt
done
