#!/bin/bash
# Where the wave cycles of the sort's three byte movers go (SQ counters), 2e9 rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_ab}
mkdir -p $OUT
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_$tag -o pmc -- python scripts/prof_sort_groupby.py sort 1 > /dev/null 2> $OUT/err_$tag.txt
  echo "== $set" >> $OUT/sort_sq.txt
  python scripts/rocprof_summary.py pmc $(find $OUT/p_$tag -name "*.db" | head -1) msd >> $OUT/sort_sq.txt 2>&1
done
find $OUT -name "*.db" -delete
cat $OUT/sort_sq.txt
