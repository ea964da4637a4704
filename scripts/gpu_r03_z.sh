#!/bin/bash
# Round 3, the very last call: bench.py once more (the CallFunction leg now times the whole Filter+Take step through Arrow).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_z}
mkdir -p $OUT
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_z/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hash_sum"]["ms"], d["sort_indices"]["ms"])
for k, v in d["other_paths"]["callfunction"].items():
    if "step" in k or k.startswith("pc."):
        print(k, v)
PY
