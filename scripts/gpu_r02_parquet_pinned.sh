#!/bin/bash
# Parquet -> HBM through the C++ binding: the column chunk read once into page-locked memory (threads, early copy) vs
# pageable staging vs the host codec; V2 and V1 data pages.  Plugin parquet tests first.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_an}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_arrow_plugin.py tests/test_parquet.py -x -q -m gpu -k "parquet or levels or snappy" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
for v in 2.0 1.0; do
  echo "== data pages V$v" | tee -a $OUT/parquet.txt
  PAGE_VERSION=$v timeout 500 python scripts/exp_parquet.py 2> $OUT/err_$v.txt | grep -v "^stats\|^file MB" | tee -a $OUT/parquet.txt
  tail -3 $OUT/err_$v.txt
done
