#!/bin/bash
# Round 3: inside aggregate_rocm's result phase (export / copies / per-column finalize), with and without kernel copies.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_r}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.txt
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--" | head -24; grep "aggregate_rocm\]" $OUT/callfunction_err.txt | sed -n '30,60p;150,175p'
