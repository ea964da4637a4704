#!/bin/bash
# Round 6 call H: the sort's level 1 write-combined by appending (sort_msd_wide_wc_form = 2) — GPU parity of the wide forms,
# sort_indices 2e9 rows with both forms on the same box, kernel trace + FETCH / WRITE counters; hash_sum after the sampling fix.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_h}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -x --durations=6 -k "sort_wide or sort_msd or config5 or groupby_lines or range_state" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 600 python bench.py --workload sort_indices --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_sort_form2.json 2> $OUT/bench_sort_form2.err; echo "bench rc=$?"; head -c 700 $OUT/bench_sort_form2.json; echo
timeout 600 python bench.py --workload sort_indices --steps 5 --warmup 2 --no-cpu-baseline --no-extras --option sort_msd_wide_wc_form=1 > $OUT/bench_sort_form1.json 2> $OUT/bench_sort_form1.err; echo "bench rc=$?"; head -c 400 $OUT/bench_sort_form1.json; echo
timeout 600 python bench.py --workload hash_sum --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_hash_sum.json 2> $OUT/bench_hash_sum.err; echo "bench rc=$?"; head -c 500 $OUT/bench_hash_sum.json; echo
RUN_TAG=${RUN_TAG:-r06_h}/prof PMC=1 WHAT=sort bash scripts/gpu_prof_sg.sh
