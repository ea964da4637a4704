#!/bin/bash
# Everything the driver runs at round end, in one call: smoke, the whole GPU suite, then the
# record run (bench + rocprofv3 kernel trace + PMC passes).  RUN_TAG names the output directory.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-record}
mkdir -p $OUT
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -20 $OUT/pytest_gpu.log
bash scripts/gpu_record_run.sh
