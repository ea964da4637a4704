#!/bin/bash
# Round 3: the gather-form filter kernel with 2 / 6 / 8 steps in flight and a 2048-row window (selectivities 10 / 25 / 50 %);
# the Acero leg with phase timing and the plan's fixed cost.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_p}
mkdir -p $OUT
export TMPDIR=/tmp
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
python scripts/exp_streams.py --tag "tree (U=4, window 1024)" 2>/dev/null | tee -a $OUT/sparse_ab.jsonl
for v in u8_w1024 u8_w2048 u6_w1024 u2_w1024; do
  cp build/variants/libarrow_amd_sparse_$v.so arrow_amd/libarrow_amd.so
  python scripts/exp_streams.py --tag "$v" 2>/dev/null | tee -a $OUT/sparse_ab.jsonl
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
python scripts/exp_streams.py --tag "tree again" 2>/dev/null | tee -a $OUT/sparse_ab.jsonl
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; grep "aggregate_rocm\]" $OUT/callfunction_err.txt | tail -45
