#!/bin/bash
# Round 6 call W: the C++ sharded entries (range-partitioned group-by, records-form sort) over the real librccl (one rank) and
# with two ranks on the one GPU; the plugin's rank / sort-key script; lint.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_w}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( time timeout 1500 python -m pytest tests/test_sharded_rccl_plugin.py tests/test_gpu_arrow_plugin.py -q -x -m gpu -k "sharded or rank_select or registered_before or order_by or table_source" --durations=6 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
