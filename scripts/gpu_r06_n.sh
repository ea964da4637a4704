#!/bin/bash
# Round 6 call N: aggregate_rocm on the range-partitioned state — the plugin's table-source test on the GPU, then the
# bench's hash_sum leg (through_acero beside the mirror) with the node's phase timing.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_n}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_arrow_plugin.py -q -x -k "table_source or acero or aggregate" ) > $OUT/pytest_plugin.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_plugin.log
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python bench.py --workload hash_sum --steps 5 --warmup 2 > $OUT/bench_hash_sum.json 2> $OUT/bench_hash_sum.err; echo "bench rc=$?"
tail -40 $OUT/bench_hash_sum.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_hash_sum.json").read().strip().splitlines()[-1])
def find(o, key):
    if isinstance(o, dict):
        if key in o: return o[key]
        for v in o.values():
            r = find(v, key)
            if r is not None: return r
    return None
print(json.dumps({"ms_per_step": d.get("ms_per_step"), "through_acero": find(d, "through_acero"), "plan": find(d, "plan")}, indent=1)[:3000])
PY
