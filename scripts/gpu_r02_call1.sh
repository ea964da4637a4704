#!/bin/bash
# Round 2, GPU call 1: (a) scatter-structure micro-benchmark, (b) kernel trace + FETCH/WRITE counters of the
# round-1 sort and group-by kernels at full size (the evidence VERDICT r1 asked for), (c) Acero morsel A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02_call1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== scatter micro-benchmark"
timeout 300 build/scatter_bench 28 > $OUT/scatter_bench.txt 2>&1; echo "rc=$?"; tail -100 $OUT/scatter_bench.txt
echo "== kernel trace: sort 2e9 + group-by 4e9"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python scripts/prof_sort_groupby.py both 2 > $OUT/prof_run.txt 2>&1; echo "rc=$?"; cat $OUT/prof_run.txt | grep -v amdgpu.ids
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/sort_groupby_kernel_stats.txt 2>&1; cat $OUT/sort_groupby_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python scripts/prof_sort_groupby.py both 1 > /dev/null 2> $OUT/pmc_$c.err; echo "pmc $c rc=$?"
  python scripts/rocprof_summary.py pmc $(find $OUT/pmc_$c -name "*.db" | head -1) >> $OUT/sort_groupby_pmc.txt 2>&1
done
cat $OUT/sort_groupby_pmc.txt
echo "== Acero morsel A/B"
timeout 300 python scripts/exp_acero_device.py > $OUT/acero_ab.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/acero_ab.txt | tail -15
find $OUT -name "*.db" -delete
du -sh $OUT
