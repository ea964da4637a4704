#!/bin/bash
# Round 6 call J: where the wave cycles of the two append scatters go (SQ counters): the sort's level 1 (2e9 rows) and the
# group-by's lines scatter (4e9 rows).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_j}
mkdir -p $OUT
export TMPDIR=/tmp
for what in sort groupby; do
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  tag=${what}_$(echo $set | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_$tag -o pmc -- python scripts/prof_sort_groupby.py $what 1 > /dev/null 2> $OUT/err_$tag.txt
  echo "== $what: $set" >> $OUT/append_scatters_sq.txt
  python scripts/rocprof_summary.py pmc $(find $OUT/p_$tag -name "*.db" | head -1) >> $OUT/append_scatters_sq.txt 2>&1
done
done
find $OUT -name "*.db" -delete
grep -E "^==|scatter1wc2|gbl_scatter|gbl_aggregate|scatter2w|bucket2w" $OUT/append_scatters_sq.txt
