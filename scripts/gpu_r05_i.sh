#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
ARROW_AMD_AGGREGATE_TIMING=1 timeout 400 python scripts/exp_acero_hash_sum_full.py 27 30 32 2>&1 | grep -v amdgpu.ids | tail -60
