#!/bin/bash
# Round 5 call A: the rec8 sort (8-byte words through the wide form) on gfx950 — its tests, the full-size configs, the
# bench line with the new parity-prefix legs, the kernel trace of the 2e9-row sort + 4e9-row group-by, rec8 on/off A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_a}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=8 -k "sort" > $OUT/pytest_sort.log 2>&1; echo "pytest sort rc=$?"; tail -14 $OUT/pytest_sort.log
timeout 400 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x > $OUT/pytest_full_size.log 2>&1; echo "pytest full size rc=$?"; tail -3 $OUT/pytest_full_size.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
p = json.load(open("$OUT/bench.json"))
print({k: p.get(k) for k in ("value", "ms_per_step", "parity", "parity_spot_check")})
print("hash_sum", p["hash_sum"].get("ms"), p["hash_sum"].get("parity_prefix"), p["hash_sum"].get("parity_plan"))
print("sort", p["sort_indices"].get("ms"), p["sort_indices"].get("parity_prefix"), p["sort_indices"].get("parity_plan"))
PY
RUN_TAG=${RUN_TAG:-r05_a}/sg WHAT=both bash scripts/gpu_prof_sg.sh
echo "== rec8 off"
ARX_OPTIONS="sort_msd_wide_rec8=0" timeout 200 python scripts/prof_sort_groupby.py sort 3 2>&1 | grep "rows run"
echo "== rec8 on, 4096-bin level 2 / other splits"
ARX_OPTIONS="sort_msd_wide_b2max=10" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run"
ARX_OPTIONS="sort_msd_wide_rpt2=24" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run"
ARX_OPTIONS="sort_msd_bucket_cpt=8" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run"
du -sh $OUT
