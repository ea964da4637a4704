#!/bin/bash
# Record run of a round: headline bench, rocprofv3 kernel trace of the same command, PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate passes) for the filter kernels + FETCH_SIZE calibration on a known byte count (selectivity 1.0
# touches every value line exactly once), then kernel trace + PMC of the sort (2e9 rows) and group-by (4e9 rows) kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-record}
mkdir -p $OUT
export TMPDIR=/tmp
echo "== headline bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
echo "== rocprofv3 kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
echo "== PMC passes (10% selectivity, the headline)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_$c.err; echo "pmc $c rc=$?"
  python scripts/rocprof_summary.py pmc $(find $OUT/pmc_$c -name "*.db" | head -1) >> $OUT/pmc_summary.txt 2>&1
done
echo "== PMC calibration: selectivity 1.0 (every 128-B value line read exactly once: 8.25 GB known)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_sparse -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --selectivity 1.0 --option filter_sparse=1 > /dev/null 2> $OUT/cal_sparse.err
echo "-- gather form (8 B/lane), selectivity 1.0" >> $OUT/pmc_summary.txt
python scripts/rocprof_summary.py pmc $(find $OUT/cal_sparse -name "*.db" | head -1) >> $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
echo "== sort 2e9 + group-by 4e9: kernel trace + PMC"
RUN_TAG=${RUN_TAG:-record}/sg PMC=1 bash scripts/gpu_prof_sg.sh
find $OUT -name "*.db" -delete
du -sh $OUT
