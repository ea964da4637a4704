#!/bin/bash
# Round 6 call Q: the group-by's scatter after the same trimming as the sort's level 1 — timed runs + kernel trace, the
# group-by parity tests, then the bench's hash_sum leg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_q}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python scripts/prof_sort_groupby.py groupby 4 2>&1 | grep "rows run\|run [0-9]" | tee $OUT/runs.txt
RUN_TAG=${RUN_TAG:-r06_q}/trace WHAT=groupby bash scripts/gpu_prof_sg.sh 2>&1 | tail -16
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupby" ) > $OUT/pytest_groupby.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_groupby.log
