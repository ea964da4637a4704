#!/bin/bash
# Round 6 call K: the whole -m gpu suite with its slowest 80 tests listed (the gate's budget: <= 500 s of the driver's 1200).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_k}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=80 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -100 $OUT/pytest_gpu.log
