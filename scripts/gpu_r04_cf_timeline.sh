#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_o}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof -o trace -- python scripts/exp_callfunction_timeline.py > $OUT/steps.txt 2>$OUT/err.txt; echo rc=$?
cat $OUT/steps.txt; tail -3 $OUT/err.txt
python scripts/rocprof_timeline.py $(find $OUT/prof -name "*.db" | head -1) 24 | tee $OUT/timeline.txt
find $OUT -name "*.db" -delete
