#!/bin/bash
# Round 3: the C++ sharded sort / group-by with one rank over the real RCCL, alone (the full run before this lost its box
# about where this test sits in the suite).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_u}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_sharded_rccl_plugin.py -x -q -m gpu > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.txt
dmesg 2>/dev/null | tail -5
