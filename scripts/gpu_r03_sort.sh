#!/bin/bash
# Round 3, sort: register-staged scatter tiles (16 / 24 rows per thread) at either level, 2^19 vs 2^20 buckets with the
# 256-thread bucket finish, end to end at 2e9 rows; kernel traces of two candidates; the plugin tests that failed in
# r03_g and the new sort parity test.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_h}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu \
  -k "register_staged or sort_wide_sampled or hash_count_min_max or int64_and_multi or dlpack" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_rpt1=24 sort_msd_wide_rpt2=8 sort_msd_wide_bits=0 sort_msd_tiny_bucket=1 sort_msd_wide_b2max=10"
timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_wide_rpt1=8" "sort_msd_wide_rpt1=16" "sort_msd_wide_rpt2=16" "sort_msd_wide_rpt2=24" \
  "sort_msd_wide_bits=20" "sort_msd_wide_bits=20 sort_msd_wide_rpt2=16" "sort_msd_wide_bits=20 sort_msd_wide_rpt2=24" \
  "sort_msd_wide_bits=20 sort_msd_tiny_bucket=0" "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16" \
  "sort_msd_wide_bits=20 sort_msd_wide_rpt1=16 sort_msd_wide_rpt2=16" "sort_msd_wide_bits=18 sort_msd_wide_rpt2=16" 2> $OUT/ab_err.txt | tee $OUT/ab.txt
tail -3 $OUT/ab_err.txt
for cfg in "" "sort_msd_wide_bits=20 sort_msd_wide_rpt2=16"; do
  tag=$(echo "${cfg:-defaults}" | tr ' =' '__')
  rm -rf /tmp/prof
  ARX_OPTIONS="$cfg" timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o sort -- python scripts/prof_sort_groupby.py sort 3 > $OUT/run_$tag.txt 2> $OUT/err_$tag.txt
  python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) msd > $OUT/sort_kernels_$tag.txt 2>&1
  echo "== $tag"; head -12 $OUT/sort_kernels_$tag.txt
done
