#!/bin/bash
# Sort of keys that share their top bits: parity tests, then 2e9 / 2^28 / 2^26 rows of keys in [0, 2^40) with the
# shared-prefix detection on and off, and the full-range case (what the detection costs when there is nothing to find).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r02_t}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
export DEFAULTS="sort_msd_prefix=1"
for rows in 2000000000 268435457 67108864; do
  RANGE_BITS=40 ROWS=$rows timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_prefix=0" 2>/dev/null | sed "s/^/keys in [0,2^40): /" | tee -a $OUT/ab.txt
done
for rows in 2000000000 67108864; do
  ROWS=$rows timeout 300 python scripts/exp_knobs.py sort "" "sort_msd_prefix=0" 2>/dev/null | sed "s/^/full-range keys:   /" | tee -a $OUT/ab.txt
done
