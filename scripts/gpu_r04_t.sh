#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_t}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_rccl_plugin.py -q -m gpu -x --timeout=400 > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.txt
timeout 900 python scripts/exp_rank_stages.py 2>&1 | grep -v amdgpu.ids | tee $OUT/rank_stages.txt
