#!/bin/bash
# Round 3: where the Acero plan's time goes (phase timing of aggregate_rocm), output batches at the input's granularity.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_n}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "key_range or table_source_rocm or acero" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; grep "aggregate_rocm" $OUT/callfunction_err.txt | tail -60
