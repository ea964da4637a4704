#!/bin/bash
# Round 3: smoke + the whole -m gpu suite on the final tree (what the driver runs at round end).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_suite}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1300 python -m pytest tests -q -m gpu -x --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $OUT/pytest_gpu.log
