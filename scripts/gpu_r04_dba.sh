#!/bin/bash
# Round 4: DELTA_BYTE_ARRAY decode on gfx950 — kernel + file tests, the plugin's encodings test, and a timing against pyarrow.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_dba}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parquet.py tests/test_gpu_arrow_plugin.py -q -m gpu -x --timeout=250 -k "delta_byte_array or delta_and_split or delta_length" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.txt
timeout 200 python scripts/exp_parquet_dba.py ${ROWS:-10000000} > $OUT/dba_timing.txt 2>&1; echo "timing rc=$?"; cat $OUT/dba_timing.txt | tail -12
