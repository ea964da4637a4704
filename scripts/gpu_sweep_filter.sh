#!/bin/bash
# Filter+Take across selectivities (the compaction form is chosen per launch from S/N).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sweep; mkdir -p $OUT
for sel in 0.01 0.1 0.25 0.26 0.5 0.9; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --selectivity $sel > $OUT/bench_$sel.json 2> $OUT/bench_$sel.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$sel.json"))
r=d["roofline"]
print("selectivity $sel: step %.3f ms  %.0f Mrows/s | filter %.3f ms alg %.2f GB -> %.0f GB/s (frac %.3f) [%s] | m2i %.3f ms | take %.3f ms (%.0f GB/s alg) | %s" % (
  d["ms_per_step"], d["value"], d["kernel_ms"]["arx_filter_exec"], r["algorithmic_bytes_per_launch"]/1e9, r["achieved"], r["frac"], r["kernel"].split()[0],
  d["kernel_ms"]["arx_mask_to_indices"], d["kernel_ms"]["arx_take"], d["kernel_ms"]["take_algorithmic_GBps"], d["parity_spot_check"]))
PY
done
