#!/bin/bash
# Round 6 call S: the whole -m gpu suite (time against the 500 s budget) after the round's additions, smoke, and the
# take-random experiment.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_s}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.txt 2>&1; tail -5 $OUT/smoke.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -36 $OUT/pytest_gpu.log
timeout 600 python scripts/exp_take_random.py > $OUT/take_random.txt 2>&1; cat $OUT/take_random.txt | tail -6
