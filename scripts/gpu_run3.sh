#!/bin/bash
# GPU pass 3 (first of this session): parity tests, headline bench, rocprof kernel trace + PMC.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/run3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(rocm-smi --showproductname 2>/dev/null | head -20; nproc; lscpu | grep "Model name"; free -g | head -2; python -c "import pyarrow as pa; print('pyarrow', pa.__version__, pa.cpu_count())") > $OUT/env.txt 2>&1
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
echo "== headline bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
ls -R $OUT/prof | head -20
python scripts/rocprof_summary.py trace $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
echo "== rocprofv3 PMC passes"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.json 2> $OUT/pmc_write.err; echo "pmc write rc=$?"
(python scripts/rocprof_summary.py pmc $(find $OUT/pmc_fetch -name "*.db" | head -1); python scripts/rocprof_summary.py pmc $(find $OUT/pmc_write -name "*.db" | head -1)) > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
