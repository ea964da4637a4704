#!/bin/bash
# Round 3: aggregate_rocm sizing its table from the key range; REE filter masks on device arrays; callfunction leg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_m}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "key_range or bytes_to_bitmap or run_end_encoded or table_source_rocm or acero or aggregate_rocm or group_by" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
timeout 900 python scripts/exp_callfunction_leg.py > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; tail -3 $OUT/callfunction_err.txt
