#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_o}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grouper or group_by_wide" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.txt
timeout 600 python scripts/exp_grouper.py > $OUT/exp_grouper.txt 2> $OUT/exp_grouper.err; echo "exp rc=$?"; cat $OUT/exp_grouper.txt; tail -5 $OUT/exp_grouper.err
