#!/bin/bash
# Round 4: DELTA_BYTE_ARRAY — the plugin's encodings test again, and the kernel trace of the timing script.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_dba2}
mkdir -p $OUT
export TMPDIR=/tmp
true
timeout 200 rocprofv3 --kernel-trace -d $OUT/t -o tr -- python scripts/exp_parquet_dba.py ${ROWS:-10000000} > $OUT/dba_timing.txt 2>&1; echo "timing rc=$?"; grep "page" $OUT/dba_timing.txt
python scripts/rocprof_summary.py trace $(find $OUT/t -name "*.db" | head -1) "" 2>&1 | head -16 | cut -c1-200 | tee $OUT/dba_trace.txt
find $OUT/t -name "*.db" -delete
