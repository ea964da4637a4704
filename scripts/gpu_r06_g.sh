#!/bin/bash
# Round 6 call G: the range-partitioned state (dense flush, K = 8, the sharded path on it) — GPU parity, the hash_sum
# leg, the kernel trace + FETCH / WRITE counters of the group-by, the virtual-rank stage table of the new sharded path.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_g}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -x --durations=4 -k "groupby_lines or groupby_range or config4 or virtual_ranks" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 600 python bench.py --workload hash_sum --steps 5 --warmup 2 --no-extras > $OUT/bench_hash_sum.json 2> $OUT/bench_hash_sum.err; echo "bench rc=$?"; head -c 2500 $OUT/bench_hash_sum.json; echo; tail -3 $OUT/bench_hash_sum.err
timeout 900 python scripts/exp_rank_stages_range.py > $OUT/virtual_rank_stage_table_range_state.txt 2> $OUT/stages.err; echo "stages rc=$?"; cat $OUT/virtual_rank_stage_table_range_state.txt; tail -3 $OUT/stages.err
RUN_TAG=${RUN_TAG:-r06_g}/prof PMC=1 WHAT=groupby bash scripts/gpu_prof_sg.sh
