#!/bin/bash
# Round 5: smoke + the whole -m gpu suite at HEAD exactly as the driver runs it (-x), then the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_suite}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1700 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1200 $OUT/bench.json; python - <<PY
import json
p = json.load(open("$OUT/bench.json"))
print({k: p.get(k) for k in ("value", "ms_per_step", "parity", "parity_spot_check")})
print("hash_sum", p["hash_sum"].get("ms"), "sort", p["sort_indices"].get("ms"))
PY
