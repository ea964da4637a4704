#!/bin/bash
# Round 6 call L: the sharded sort's records form on gfx950 (arx_sort_records parity, the virtual-rank stage table), the
# plugin scripts the round touched, and the WHOLE bench line (config-3 CPU baselines, CallFunction timings, the default-state
# Acero plans).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_l}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py tests/test_sharded_rccl_plugin.py -q -m gpu -x --durations=6 -k "sort_records or sort_virtual or stock_acero or acero_plan_over or table_source_rocm_delivers or decimal or group_by_wide_and_multiple or world2_on_one_gpu" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 900 python scripts/exp_rank_stages_sort_records.py > $OUT/virtual_rank_stage_table_sort_records.txt 2> $OUT/stages.err; echo "stages rc=$?"; cat $OUT/virtual_rank_stage_table_sort_records.txt; tail -3 $OUT/stages.err
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_l/bench.json".replace("r06_l", __import__("os").environ.get("RUN_TAG","r06_l"))))
print({k: d[k] for k in ("value","ms_per_step")}, d["roofline"])
op=d.get("other_paths",{})
for k in ("cast_f64_f32","greater_f64"): print(k, json.dumps(op.get(k))[:900])
print("hash_sum", {k: d["hash_sum"].get(k) for k in ("ms","plan","checksum_matches_sum_of_values","parity_prefix")}, d["hash_sum"].get("through_acero"))
print("sort", {k: d["sort_indices"].get(k) for k in ("ms","parity_prefix")})
cf=op.get("callfunction",{})
for k,v in cf.items():
    if "acero" in k: print(k, v)
PY
tail -3 $OUT/bench.err
