#!/bin/bash
# Round 5 call N: how much of a rank's merge / finalize time is the size of the owner's table (2 x records vs 4 x one block)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_n}
mkdir -p $OUT
export TMPDIR=/tmp
for f in 16 4 3; do
  echo "owner table = $f x one block"; OWNED_FACTOR=$f timeout 300 python scripts/exp_rank_profile.py 2>&1 | grep "^rep" | tee -a $OUT/owned_factor_$f.log
done
