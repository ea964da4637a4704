#!/bin/bash
# Round 5 call L: the sharded group-by's local pass without the local table (arx_groupby_sum_i64_consume_partials) —
# device parity (kernel tier, the C++ sharded entry over the real librccl and with two ranks on one GPU), then the
# virtual-rank stage table with both forms of the local pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_l}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_rccl_plugin.py -q -m gpu -x --durations=6 -k "consume_partials or virtual_ranks or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 900 python scripts/exp_rank_stages.py > $OUT/virtual_rank_stage_table.txt 2> $OUT/stages.err; echo "stages rc=$?"; cat $OUT/virtual_rank_stage_table.txt; tail -3 $OUT/stages.err
