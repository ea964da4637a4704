#!/bin/bash
# Round 3: one 64-bit atomic per PAIR of bins in the scatters (sort levels 1 / 2, group-by flat level); block-level
# reduction in front of single-address atomics (bytes_to_bitmap, popcount, key range).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_s}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort_wide or sort_keys_with or register_staged or groupby_wide or groupby_probe or key_range or bytes_to_bitmap or null_count or filter" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_wide_rpt1=24 sort_msd_wide_rpt2=16 sort_msd_wide_bits=0 sort_msd_tiny_bucket=2 sort_msd_wide_b2max=11"
timeout 600 python scripts/exp_knobs.py sort "" "sort_msd_wide_rpt2=8" "sort_msd_wide_b2max=10 sort_msd_wide_rpt2=8" "sort_msd_wide_b2max=10" "sort_msd_wide_rpt1=16" "sort_msd_wide_bits=19 sort_msd_wide_b2max=10 sort_msd_wide_rpt2=8" 2> $OUT/ab_err.txt | tee $OUT/ab.txt
DEFAULTS="" timeout 300 python scripts/exp_knobs.py groupby "" 2>/dev/null | tee -a $OUT/ab.txt
rm -rf /tmp/prof
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o both -- python scripts/prof_sort_groupby.py both 2 > $OUT/run.txt 2> $OUT/err.txt
python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) > $OUT/kernels.txt 2>&1; head -14 $OUT/kernels.txt
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused\|pc.filter" $OUT/callfunction.json | grep -v "^--" | head -40; grep "aggregate_rocm\]" $OUT/callfunction_err.txt | sed -n '30,45p'
