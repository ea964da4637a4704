#!/bin/bash
# Round 4: a slice of the -m gpu suite (TESTS = pytest arguments), -x, per-test wall-clock limit; used to run the
# suite in pieces after a box was lost under the whole-suite call (which piece was running cannot be told otherwise).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_tests}
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SMOKE:-0}" = 1 ]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
fi
timeout ${LIMIT:-900} python -m pytest ${TESTS:-tests} -q -m gpu -x --timeout=${PER_TEST:-400} --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -${TAIL:-18} $OUT/pytest_gpu.log
free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
