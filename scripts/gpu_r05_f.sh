#!/bin/bash
# Round 5 call F: level 2 of the rec8 sort in small workgroups (A/B of the shapes), kernel trace, rec8 tests, the
# headline through CallFunction with the fused null count.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_f}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
RUN_TAG=${RUN_TAG:-r05_f}/sg WHAT=sort bash scripts/gpu_prof_sg.sh
for o in "sort_msd_wide_l2w=0" "sort_msd_wide_l2w=2" "sort_msd_wide_l2w=3" "sort_msd_wide_l2w=1 sort_msd_wide_b2max=10" "sort_msd_wide_l2w=3 sort_msd_wide_b2max=10"; do
  echo "== $o"
  ARX_OPTIONS="$o" timeout 200 python scripts/prof_sort_groupby.py sort 2 2>&1 | grep "rows run [12]"
done
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=5 -k "sort_wide_rec8 or filter" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -9 $OUT/pytest.log
timeout 300 python bench.py --no-extras > $OUT/bench_noextras.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
p = json.load(open("$OUT/bench_noextras.json"))
print({k: p.get(k) for k in ("value", "ms_per_step", "python_mirror", "kernel_ms", "parity_spot_check", "callfunction_results_equal_the_mirrors")})
PY
