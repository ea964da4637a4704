#!/bin/bash
# Round 3: the group-by's flat level with one 64-bit cursor atomic per pair of bins against one per bin, interleaved,
# end to end and under the kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_t}
mkdir -p $OUT
export TMPDIR=/tmp
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
for rep in 1 2; do
  for v in pair single; do
    if [ $v = pair ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_gbpair0.so arrow_amd/libarrow_amd.so; fi
    echo "== $v rep $rep" | tee -a $OUT/gb_pair_ab.txt
    DEFAULTS="" timeout 300 python scripts/exp_knobs.py groupby "" 2>/dev/null | tee -a $OUT/gb_pair_ab.txt
  done
done
for v in single pair; do
  if [ $v = pair ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_gbpair0.so arrow_amd/libarrow_amd.so; fi
  rm -rf /tmp/prof
  timeout 400 rocprofv3 --kernel-trace -d /tmp/prof -o gb -- python scripts/prof_sort_groupby.py groupby 2 > $OUT/run_$v.txt 2> $OUT/err_$v.txt
  echo "== kernel trace, $v" | tee -a $OUT/gb_pair_ab.txt
  grep "rows run" $OUT/run_$v.txt | tee -a $OUT/gb_pair_ab.txt
  python scripts/rocprof_summary.py trace $(find /tmp/prof -name "*.db" | head -1) gbp > $OUT/kernels_$v.txt 2>&1; head -4 $OUT/kernels_$v.txt | tee -a $OUT/gb_pair_ab.txt
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
