#!/bin/bash
# Round 6 call P: A/B of level 1's flush (per-wave compaction vs the queue of full lines) with a kernel trace of each,
# on one box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_p}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for cfg in ${CFGS:-"sort_msd_wide_wc_typed=1" "sort_msd_wide_wc_typed=2"}; do
  tag=$(echo $cfg | tr '=,' '__')
  echo "== $cfg" | tee -a $OUT/ab.txt
  ARX_OPTIONS="$(echo $cfg | tr ',' ' ')" timeout 300 python scripts/prof_sort_groupby.py sort 4 2>&1 | grep "rows run" | tee -a $OUT/ab.txt
  ARX_OPTIONS="$(echo $cfg | tr ',' ' ')" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o trace -- python scripts/prof_sort_groupby.py sort 3 > $OUT/run_$tag.txt 2>&1
  python scripts/rocprof_summary.py trace $(find $OUT/prof_$tag -name "*.db" | head -1) 2>&1 | head -8 | cut -c1-200 | tee -a $OUT/ab.txt
done
find $OUT -name "*.db" -delete
