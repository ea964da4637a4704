#!/bin/bash
# A/B of tuning options on the headline bench: bash scripts/gpu_ab.sh "filter_sparse=0" "filter_sparse=1" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/ab; mkdir -p $OUT
export TMPDIR=/tmp
for opt in "$@"; do
  tag=$(echo "$opt" | tr ' =' '__')
  args=""; for kv in $opt; do [ "$kv" = "none" ] || args="$args --option $kv"; done
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $args > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$tag.json"))
print("$opt", "value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["kernel_ms"], d["parity_spot_check"])
PY
done
