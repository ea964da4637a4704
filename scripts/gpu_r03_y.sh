#!/bin/bash
# Round 3: the late additions on gfx950 — hash_any / hash_all (C ABI parity + under the stock GroupByNode), coalesce_rocm's
# copying path, REE golden vectors, the C++ sharded sort.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_y}
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py tests/test_sharded_rccl_plugin.py -x -q -m gpu -k "hash_any_all or hash_count_min_max or table_source_rocm or run_end_encoded or sharded or int64_and_multi or scalar_aggregates" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.txt
