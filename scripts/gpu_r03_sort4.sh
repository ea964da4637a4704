#!/bin/bash
# Round 3: 2^20 buckets (rule lg - 11) with the 9-rows-per-thread finish (five per CU), at 2e9 and 2^28 rows; the
# callfunction leg (table_source_rocm); the gather cache-policy experiment of take.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_k}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register_staged or sort_wide_sampled or sort_keys_with" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_rpt1=24 sort_msd_wide_rpt2=8 sort_msd_wide_bits=0 sort_msd_tiny_bucket=1 sort_msd_wide_b2max=10 sort_msd_bucket_cpt=4"
NEW="sort_msd_wide_b2max=11 sort_msd_wide_rpt2=16"
timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_wide_bits=20 $NEW" "sort_msd_wide_bits=20 $NEW sort_msd_tiny_bucket=2" "sort_msd_wide_bits=20 $NEW sort_msd_tiny_bucket=2 sort_msd_bucket_cpt=8" \
  "sort_msd_wide_bits=20 $NEW sort_msd_bucket_cpt=8" 2> $OUT/ab_err.txt | tee $OUT/ab.txt
ROWS=268435457 timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_wide_bits=17 $NEW" "sort_msd_wide_bits=17 $NEW sort_msd_tiny_bucket=2" "sort_msd_wide_bits=17 sort_msd_wide_b2max=9 sort_msd_tiny_bucket=2" \
  "sort_msd_wide_bits=17 sort_msd_wide_b2max=10 sort_msd_tiny_bucket=2" "sort_msd_wide_bits=17 sort_msd_wide_b2max=10 sort_msd_wide_rpt2=16 sort_msd_tiny_bucket=2" 2>> $OUT/ab_err.txt | tee -a $OUT/ab.txt
timeout 900 python scripts/exp_callfunction_leg.py > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; tail -3 $OUT/callfunction_err.txt
RUN_TAG=${RUN_TAG:-r03_k} bash scripts/gpu_r03_take.sh
