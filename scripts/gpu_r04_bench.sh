#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_p}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
keep = {k: d[k] for k in ("metric", "value", "ms_per_step", "api", "python_mirror", "roofline", "kernel_ms", "parity_spot_check") if k in d}
for k in ("api_error_fell_back_to_the_mirror", "callfunction_results_equal_the_mirrors"):
    if k in d: keep[k] = d[k]
keep["cpu_baseline"] = {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "cores", "sample", "single_thread_mrows_per_s", "acero_mrows_per_s")}
for leg in ("hash_sum", "sort_indices"):
    if leg in d: keep[leg] = {k: d[leg].get(k) for k in ("ms", "mrows_per_s", "error")}; keep[leg]["frac"] = d[leg].get("roofline", {}).get("frac"); keep[leg]["cpu"] = d[leg].get("cpu_baseline", {}).get("value")
cf = d.get("other_paths", {}).get("callfunction", {})
keep["callfunction_step"] = {k: v for k, v in cf.items() if "Filter+Take" in k or "error" in k}
sec = d.get("other_paths", {}).get("secondary_configs", {})
keep["secondary"] = {k: (v.get("ms"), v.get("roofline_frac")) for k, v in sec.items() if isinstance(v, dict)}
print(json.dumps(keep, indent=1))
PY
tail -5 $OUT/bench.err
