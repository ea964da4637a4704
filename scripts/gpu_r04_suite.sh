#!/bin/bash
# Round 4 call 1: smoke + the whole -m gpu suite at HEAD exactly as the driver runs it (-x), then the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_suite}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
