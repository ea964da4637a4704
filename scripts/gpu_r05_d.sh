#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_d}
mkdir -p $OUT
for args in "20000003 18 1 4 24 16 9 256 1" "5000003 14 0 0 8 16 5 64 1" "20000003 18 0 0 16 16 9 256 1"; do
  echo "== $args"
  timeout 200 python scripts/exp_sort_rec8_case.py $args 2>&1 | grep -v "^  File\|Extension modules\|^$" | tail -4
done
