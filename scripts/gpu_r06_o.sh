#!/bin/bash
# Round 6 call O: level 1 of the sort compiled for its key type (sort_msd_wide_wc_typed) — timed runs A/B, then the
# kernel trace of the default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_o}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for cfg in "sort_msd_wide_wc_typed=1" "sort_msd_wide_wc_typed=0" ${EXTRA_CFGS:-}; do
  echo "== $cfg" | tee -a $OUT/ab.txt
  ARX_OPTIONS="$(echo $cfg | tr ',' ' ')" timeout 300 python scripts/prof_sort_groupby.py sort 4 2>&1 | grep "rows run" | tee -a $OUT/ab.txt
done
RUN_TAG=${RUN_TAG:-r06_o}/trace WHAT=sort bash scripts/gpu_prof_sg.sh 2>&1 | tail -25
