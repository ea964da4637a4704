#!/bin/bash
# Round 4: is the memory pipe what holds the filter compaction at 10 / 25 / 50 % selectivity?  L2 -> memory
# credit stalls, L2 request counts and CU -> L2 requests of compact_sparse_kernel, one PMC set per pass (the same sets on a
# streaming copy: profiles/r04_w_*, r04_y_*).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_filter}
mkdir -p $OUT
export TMPDIR=/tmp
for sel in 0.10 0.25 0.50; do
  for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $set | cut -d" " -f1)
    timeout 150 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_${sel}_$tag -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --selectivity $sel > /dev/null 2> $OUT/err_${sel}_$tag.txt
    echo "pass $sel $tag rc=$?"
    echo "== selectivity $sel: $set" >> $OUT/filter_mem.txt
    python scripts/rocprof_summary.py pmc $(find $OUT/p_${sel}_$tag -name "*.db" | head -1) compact_sparse_kernel >> $OUT/filter_mem.txt 2>&1
    find $OUT/p_${sel}_$tag -name "*.db" -delete
  done
done
cut -c1-40,88-170 $OUT/filter_mem.txt
