#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05_k
ulimit -c 0
timeout 500 python scripts/exp_rank_stages.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_k/virtual_rank_stage_table.txt | tail -12
