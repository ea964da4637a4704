#!/bin/bash
# First GPU call of the next round: everything that has only run on the SIMT emulator so far
# (tests/test_zz_gpu_first_run.py + the Parquet GPU tests), WITHOUT -x so that one failure does not hide
# the others, then the Parquet end-to-end timing.  ~3-4 GPU minutes.  RUN_TAG names the output directory.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-first_run}
mkdir -p $OUT
echo "== never-run-on-GPU tests"
timeout 1200 python -m pytest tests/test_zz_gpu_first_run.py tests/test_parquet.py -q -m gpu > $OUT/pytest_first_run.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_first_run.log
echo "== Parquet end to end (host prep vs device decode vs pyarrow)"
timeout 600 python scripts/exp_parquet.py > $OUT/exp_parquet.log 2>&1; echo "exp_parquet rc=$?"; tail -8 $OUT/exp_parquet.log
echo "== throughput of the kernels that have never been timed"
timeout 600 python scripts/exp_new_kernels.py > $OUT/exp_new_kernels.log 2>&1; echo "exp_new_kernels rc=$?"; tail -10 $OUT/exp_new_kernels.log
