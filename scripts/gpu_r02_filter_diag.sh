#!/bin/bash
# Where the wave cycles of the filter compaction go at 10 / 25 / 50 % selectivity (SQ counters, one pass each).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_u}
mkdir -p $OUT
export TMPDIR=/tmp
for sel in 0.10 0.25 0.50; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU"; do
    tag=$(echo $set | cut -d" " -f1)
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_${sel}_$tag -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --selectivity $sel > /dev/null 2> $OUT/err_${sel}_$tag.txt
    echo "== selectivity $sel: $set" >> $OUT/filter_sq.txt
    python scripts/rocprof_summary.py pmc $(find $OUT/p_${sel}_$tag -name "*.db" | head -1) compact_sparse_kernel >> $OUT/filter_sq.txt 2>&1
  done
done
find $OUT -name "*.db" -delete
cat $OUT/filter_sq.txt
