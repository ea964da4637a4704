#!/bin/bash
# Round 3: the device Grouper's GPU tests one at a time, smallest first, each in its own process under a wall-clock
# kill; the log is flushed after every test so that a lost box still tells where it happened.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_grouper
mkdir -p $OUT
export ARROW_AMD_GPU_GROUPER=1
ids=$(python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grouper" --collect-only 2>/dev/null | grep "::")
echo "$ids" > $OUT/collected.txt
for t in $ids; do
  echo "=== $t" >> $OUT/log.txt; sync
  timeout 150 python -m pytest "$t" -x -q -m gpu > $OUT/one.txt 2>&1; rc=$?
  tail -3 $OUT/one.txt >> $OUT/log.txt; echo "rc=$rc" >> $OUT/log.txt; sync
  echo "$t rc=$rc"
  if [ $rc -ne 0 ]; then cat $OUT/one.txt | tail -40; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; break; fi
done
echo "== exp_grouper (2^26 rows)"
ROWS=67108864 GROUPS=2000000 timeout 300 python scripts/exp_grouper.py > $OUT/exp_grouper.txt 2> $OUT/exp_grouper.err; echo "exp rc=$?"; cat $OUT/exp_grouper.txt; tail -3 $OUT/exp_grouper.err
