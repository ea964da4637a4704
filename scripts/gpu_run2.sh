#!/bin/bash
# GPU pass 2: parity tests (+ Arrow plugin), A/B of the filter variants, headline bench, rocprof + PMC.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/run2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
echo "== filter variants A/B (1B rows)"
for b in 4 1; do for p in 1 0; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras \
     --option filter_batch=$b --option filter_pipe=$p > $OUT/bench_b${b}_p${p}.json 2> $OUT/bench_b${b}_p${p}.err
  echo "batch=$b pipe=$p rc=$?"; python -c "
import json,sys
d=json.load(open('$OUT/bench_b${b}_p${p}.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms'])"
done; done
echo "== headline bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
echo "== rocprofv3 kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
echo "== rocprofv3 PMC passes"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.json 2> $OUT/pmc_write.err; echo "pmc write rc=$?"
(python scripts/rocprof_summary.py pmc $(find $OUT/pmc_fetch -name "*.db" | head -1); python scripts/rocprof_summary.py pmc $(find $OUT/pmc_write -name "*.db" | head -1)) > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
find $OUT -name "*.db" -delete
du -sh $OUT
