#!/bin/bash
# Round 4: group-by / sort library variants under the kernel trace (end-to-end times of separate processes move by
# +-3 ms): per variant the big kernels' average durations.  VARIANTS = names under build/variants (tree = the tree's lib).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_h}
mkdir -p $OUT
export TMPDIR=/tmp
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
W=${WHAT:-groupby}
for v in ${VARIANTS:-tree}; do
  if [ $v = tree ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_$v.so arrow_amd/libarrow_amd.so; fi
  echo "== $v" | tee -a $OUT/ab_trace.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o trace -- python scripts/prof_sort_groupby.py $W ${RUNS:-3} 2>/dev/null | grep "run [1-9]" | tee -a $OUT/ab_trace.txt
  python scripts/rocprof_summary.py trace $(find $OUT/prof_$v -name "*.db" | head -1) 2>&1 | head -${HEAD:-5} | cut -c1-60,88-170 | tee -a $OUT/ab_trace.txt
  find $OUT/prof_$v -name "*.db" -delete
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
