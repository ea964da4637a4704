#!/bin/bash
# Round 5 call J: the cases added after the whole-suite run (string / decimal sort keys, multi-chunk sort, the span consume,
# the new grouped aggregates, kernel-tier product / edge rows), then the bench line with hash_sum through Acero.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_j}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_arrow_plugin.py tests/test_gpu_parity.py -q -m gpu -x --durations=6 -k "first_last or stock_group_by or decimal or filter_and_take_of_device_resident_batches or order_by or table_source or general or hash_product_group_edge or acero" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
p = json.load(open("$OUT/bench.json"))
print({k: p.get(k) for k in ("value", "ms_per_step")}, p["parity"]["hash_sum_prefix"], p["parity"]["sort_prefix"])
print("hash_sum", p["hash_sum"].get("ms"), p["hash_sum"].get("through_acero"))
print("sort", p["sort_indices"].get("ms"))
PY
