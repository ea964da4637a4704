#!/bin/bash
# Round 4: grouped float sums on gfx950 — kernel parity, the plugin test, a timing against the reference's GroupByNode.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_fsum}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -q -m gpu -x --timeout=300 -k "grouped_float_sum or row_order_sums or hash_min_max_of_floats" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.txt
timeout 250 python scripts/exp_float_groupby.py > $OUT/float_groupby_timing.txt 2>&1; echo "timing rc=$?"; tail -2 $OUT/float_groupby_timing.txt
timeout 250 python scripts/exp_float_groupby.py 67108864 100 >> $OUT/float_groupby_timing.txt 2>&1; echo "timing rc=$?"; tail -1 $OUT/float_groupby_timing.txt
