#!/bin/bash
# Round 3: the bucket finish with the lighter ranking loop (keys only, two per step) and 4 / 8 sub-bucket counters per
# thread; table_source_rocm against table_source in front of filter -> project -> aggregate_rocm.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_j}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "register_staged or sort_wide_sampled or sort_keys_with or table_source_rocm or sort_msd" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_rpt1=24 sort_msd_wide_rpt2=8 sort_msd_wide_bits=0 sort_msd_tiny_bucket=1 sort_msd_wide_b2max=10 sort_msd_bucket_cpt=4 sort_msd_final_rows_log2=1"
timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_bucket_cpt=8" "sort_msd_final_rows_log2=2" \
  "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16" "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16 sort_msd_bucket_cpt=8" \
  "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16 sort_msd_final_rows_log2=2" 2> $OUT/ab_err.txt | tee $OUT/ab.txt
tail -2 $OUT/ab_err.txt
for cfg in "" "sort_msd_wide_bits=20 sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16"; do
  tag=$(echo "${cfg:-defaults}" | tr ' =' '__')
  rm -rf /tmp/prof
  ARX_OPTIONS="$cfg" timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o sort -- python scripts/prof_sort_groupby.py sort 3 > $OUT/run_$tag.txt 2> $OUT/err_$tag.txt
  python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) msd > $OUT/sort_kernels_$tag.txt 2>&1
  echo "== $tag"; head -5 $OUT/sort_kernels_$tag.txt
done
timeout 900 python scripts/exp_callfunction_leg.py > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; tail -3 $OUT/callfunction_err.txt
