#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sweep; mkdir -p $OUT
for sel in 0.25 0.5 0.9 1.0; do for mode in 0 1; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --selectivity $sel --option filter_sparse=$mode > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.load(open("$OUT/b.json"))
print("selectivity $sel sparse=$mode: filter %.3f ms frac %.3f %s" % (d["kernel_ms"]["arx_filter_exec"], d["roofline"]["frac"], d["parity_spot_check"]))
PY
done; done
