#!/bin/bash
# Round 3, last call: the record run again on the final library (wide record loads in the group-by's aggregate, 2^25-row
# probe), then the group-by GPU tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
RUN_TAG=${RUN_TAG:-r03_final2} bash scripts/gpu_record_run.sh > gpurun_out/.record.log 2>&1; tail -3 gpurun_out/.record.log
OUT=gpurun_out/${RUN_TAG:-r03_final2}
head -c 600 $OUT/bench.json; echo
timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groupby or group_by or hash" --deselect "tests/test_gpu_parity.py::test_group_by_wide_and_multiple_keys" > $OUT/tests_groupby.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests_groupby.txt
