#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_fsum_trace}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d $OUT/t -o tr -- python scripts/exp_float_groupby.py ${FARGS:-67108864 100} > $OUT/timing.txt 2>&1; tail -1 $OUT/timing.txt
python scripts/rocprof_summary.py trace $(find $OUT/t -name "*.db" | head -1) "" 2>&1 | head -14 | cut -c1-190 | tee $OUT/trace.txt
find $OUT/t -name "*.db" -delete
