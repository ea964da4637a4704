#!/bin/bash
# First GPU pass: smoke, parity tests, A/B of the filter variants, headline bench, rocprof.
# Usage on the GPU box (via gpurun): bash scripts/gpu_run1.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/run1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" | tee $OUT/env.txt
(rocm-smi --showproductname 2>/dev/null | head -20; nproc; lscpu | grep "Model name"; free -g | head -2; python -c "import pyarrow as pa; print('pyarrow', pa.__version__, pa.cpu_count())") >> $OUT/env.txt 2>&1

echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log

echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log

echo "== filter variants A/B (1B rows)"
for b in 1 4; do for d in 0 1; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras \
     --option filter_batch=$b --option filter_dense=$d > $OUT/bench_b${b}_d${d}.json 2> $OUT/bench_b${b}_d${d}.err
  echo "batch=$b dense=$d rc=$?"; cat $OUT/bench_b${b}_d${d}.json
done; done

echo "== headline bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json

echo "== rocprofv3 kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -20 $f; done
# keep the merged output small: drop the raw per-dispatch traces (can be 10s of MB)
find $OUT/prof -name "*kernel_trace*.csv" -size +5M -delete

echo "== rocprofv3 PMC passes (own runs, no tracing domains besides kernel-trace)"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.json 2> $OUT/pmc_write.err; echo "pmc write rc=$?"
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
