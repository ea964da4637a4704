#!/bin/bash
# Round 5 call P: the grouped moments (hash_variance / hash_stddev / hash_skew / hash_kurtosis) on gfx950 — kernel tier
# against the oracle, the plugin case against the stock GroupByNode — and the direct local pass once more at this HEAD.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r05_p}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -q -m gpu -x --durations=6 -k "group_moments or variance or consume_partials or first_last" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
