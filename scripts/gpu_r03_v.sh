#!/bin/bash
# Round 3: the sweeping (dense) filter form against the gather form at 25 / 50 % selectivity, now that both carry
# non-temporal accesses (round 1 measured the gather form faster at every selectivity for 8-byte values).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_v}
mkdir -p $OUT
export TMPDIR=/tmp
for opt in "filter_sparse=-1" "filter_sparse=0" "filter_sparse=-1" "filter_sparse=0"; do
  for sel in 0.25 0.5 0.1; do
    echo "== $opt selectivity $sel" | tee -a $OUT/filter_forms.txt
    timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --selectivity $sel --option $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('filter ms', d['kernel_ms']['arx_filter_exec'], 'frac', d['roofline']['frac'], 'kernel', d['roofline']['kernel'])" | tee -a $OUT/filter_forms.txt
  done
done
