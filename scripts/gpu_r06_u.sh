#!/bin/bash
# Round 6 call U: the sharded sort's receiver reading the records in place — the virtual-rank stage table again (P = 1, 2, 4, 8),
# the records tests on gfx950, and the kernel trace of one rank of 4.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_u}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sort_records" ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python scripts/exp_rank_stages_sort_records.py 2>&1 | grep -v "^W2026" | tee $OUT/stage_table.txt
WORLDS=4 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python scripts/exp_rank_stages_sort_records.py > $OUT/run.txt 2>&1; echo "rc=$?"
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -12 $OUT/kernel_stats.txt | cut -c1-200
find $OUT -name "*.db" -delete
