#!/bin/bash
# Round 5 call H: the plugin cases added / changed late in the round on gfx950, then the bench line with hash_sum through Acero.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_h}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_arrow_plugin.py -q -m gpu -x --durations=6 -k "first_last or stock_group_by or decimal or filter_and_take_of_device_resident_batches or order_by or golden_grouped or general or count_distinct" > $OUT/pytest_plugin.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_plugin.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
p = json.load(open("$OUT/bench.json"))
print({k: p.get(k) for k in ("value", "ms_per_step", "parity")})
print("hash_sum", p["hash_sum"].get("ms"), p["hash_sum"].get("through_acero"))
print("sort", p["sort_indices"].get("ms"), p["sort_indices"]["roofline"])
PY
tail -3 $OUT/bench.err
