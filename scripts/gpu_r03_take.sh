#!/bin/bash
# Round 3, take: which cache policy of the gather load fetches less than a 128-B line per 8-B value?  Timing of every
# policy (monotonic 10 % density / random indices), then FETCH_SIZE per policy kernel in separate passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_k}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 build/gather_policy_bench 30 100000000 both | tee $OUT/gather_policy_times.txt
for which in mono random; do
  rm -rf /tmp/pmc_$which
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_$which -o pmc -- build/gather_policy_bench 30 100000000 $which > /dev/null 2> $OUT/pmc_$which.err; echo "pmc $which rc=$?"
  echo "== FETCH_SIZE (KiB at 64 B per request: x2 on gfx950), $which indices" >> $OUT/gather_policy_fetch.txt
  python scripts/rocprof_summary.py pmc $(find /tmp/pmc_$which -name "*.db" | head -1) gather_policy >> $OUT/gather_policy_fetch.txt 2>&1
done
cat $OUT/gather_policy_fetch.txt
