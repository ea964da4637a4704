#!/bin/bash
# Round 3: the callfunction leg with pinned result buffers / device-packed validity in aggregate_rocm; the sort in its
# new default form (2^20 buckets, 24 / 16-row tiles, 256-thread finish): kernel trace, FETCH / WRITE passes, LDS counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_l}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "bytes_to_bitmap or table_source_rocm or acero or aggregate_rocm or register_staged" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
timeout 900 python scripts/exp_callfunction_leg.py > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; tail -3 $OUT/callfunction_err.txt
RUN_TAG=${RUN_TAG:-r03_l} PMC=1 WHAT=sort bash scripts/gpu_prof_sg.sh
set="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_lds -o pmc -- python scripts/prof_sort_groupby.py sort 1 > /dev/null 2> $OUT/err_lds.txt
echo "== $set" > $OUT/sort_lds_counters.txt
python scripts/rocprof_summary.py pmc $(find $OUT/p_lds -name "*.db" | head -1) msd >> $OUT/sort_lds_counters.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/sort_lds_counters.txt
