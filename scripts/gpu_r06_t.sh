#!/bin/bash
# Round 6 call T: kernel trace of one virtual rank of 4 of the sharded sort's records form (5e8 records through
# arx_sort_records) — where its 13 ms go.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_t}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
WORLDS=${WORLDS:-4} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python scripts/exp_rank_stages_sort_records.py > $OUT/run.txt 2>&1; echo "rc=$?"
cat $OUT/run.txt | tail -4
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt | cut -c1-200
find $OUT -name "*.db" -delete
