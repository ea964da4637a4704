#!/bin/bash
# Kernel trace (and optionally FETCH/WRITE counters: PMC=1) of configs[4] sort + configs[3] group-by at full size.
# usage: RUN_TAG=name [PMC=1] [WHAT=both|sort|groupby] [ARX_OPTIONS="k=v ..."] bash scripts/gpu_prof_sg.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-prof_sg}
mkdir -p $OUT
export TMPDIR=/tmp
W=${WHAT:-both}
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python scripts/prof_sort_groupby.py $W 2 > $OUT/run.txt 2>&1; echo "rc=$?"
grep -E "rows run|Error|error" $OUT/run.txt
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
if [ "${PMC:-0}" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python scripts/prof_sort_groupby.py $W 1 > /dev/null 2> $OUT/pmc_$c.err; echo "pmc $c rc=$?"
    python scripts/rocprof_summary.py pmc $(find $OUT/pmc_$c -name "*.db" | head -1) >> $OUT/pmc.txt 2>&1
  done
  cat $OUT/pmc.txt
fi
find $OUT -name "*.db" -delete
