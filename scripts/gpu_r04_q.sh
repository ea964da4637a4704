#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_q}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_group_keys.py tests/test_gpu_arrow_plugin.py tests/test_gpu_parity.py -q -m gpu -x --timeout=400 -k "utf8 or binary_key or wrap_device or fill_null or float or coalesce or hash_minmax or vector_hash or unique or groupby_wide or group_by_sum or hash_sum" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.txt
timeout 600 python scripts/exp_string_keys.py 2>&1 | grep -v amdgpu.ids | tee $OUT/string_keys.txt
