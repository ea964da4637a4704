#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_r}
mkdir -p $OUT
timeout 600 python scripts/exp_string_keys.py 2>&1 | grep -v amdgpu.ids | tee $OUT/string_keys.txt
