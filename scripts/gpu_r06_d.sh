#!/bin/bash
# Round 6 call D: the lines plan inside arx_groupby_sum_i64_consume — parity on gfx950 (kernel tier + the 1e9-row config),
# then the hash_sum leg of the bench with the plan on and off on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_d}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -x --durations=8 -k "groupby or hash_sum or config4" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 600 python bench.py --workload hash_sum --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_hash_sum_lines.json 2> $OUT/bench_hash_sum_lines.err; echo "bench rc=$?"; head -c 1200 $OUT/bench_hash_sum_lines.json; echo
timeout 600 python bench.py --workload hash_sum --steps 5 --warmup 2 --no-cpu-baseline --no-extras --option groupby_lines=0 > $OUT/bench_hash_sum_wide.json 2> $OUT/bench_hash_sum_wide.err; echo "bench rc=$?"; head -c 400 $OUT/bench_hash_sum_wide.json; echo
