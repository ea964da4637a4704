#!/bin/bash
# Round 4: fixed_size_list / list selection on gfx950 — kernel parity, the plugin test, an embeddings timing.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_nested}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -q -m gpu -x --timeout=300 -k "rows_of_any_width or fixed_size_list or large_utf8" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.txt
timeout 200 python scripts/exp_fsl_take.py > $OUT/fsl_timing.txt 2>&1; echo "timing rc=$?"; tail -4 $OUT/fsl_timing.txt
timeout 200 python scripts/exp_fsl_take.py 16777216 4 >> $OUT/fsl_timing.txt 2>&1; echo "timing rc=$?"; tail -3 $OUT/fsl_timing.txt
