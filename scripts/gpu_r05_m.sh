#!/bin/bash
# Round 5 call M: consume_partials device parity on the rebuilt library, then a kernel trace of one virtual rank at P = 8.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_m}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=6 -k "consume_partials or virtual_ranks" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
for d in 1 0; do
  rm -rf /tmp/prof_$d
  (cd /tmp && DIRECT=$d timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$d -o rank -- python $OLDPWD/scripts/exp_rank_profile.py) > $OUT/rank_direct$d.log 2>&1
  grep "^rep" $OUT/rank_direct$d.log
  db=$(find /tmp/prof_$d -name "*.db" | head -1)
  python scripts/rocprof_summary.py trace "$db" "" > $OUT/rank_direct${d}_kernels.txt 2>&1
  head -24 $OUT/rank_direct${d}_kernels.txt | cut -c1-200
done
