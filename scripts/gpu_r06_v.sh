#!/bin/bash
# Round 6 call V: the sharded sort with the key window and the splitters from a sample — stage table (P = 1, 2, 4, 8), the
# virtual-rank and records tests on gfx950.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_v}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sort_records or virtual_ranks" ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python scripts/exp_rank_stages_sort_records.py 2>&1 | grep -v "^W2026" | tee $OUT/stage_table.txt
