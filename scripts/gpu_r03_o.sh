#!/bin/bash
# Round 3: aggregate_rocm with its result kept in HBM; cache policy of the group-by's wide-form streams (kGbNt variants).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_o}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "key_range or table_source_rocm or run_end_encoded" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; tail -2 $OUT/callfunction_err.txt
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
echo "== tree (kGbNt=0)" | tee -a $OUT/gb_nt_ab.txt
timeout 300 python scripts/exp_knobs.py groupby "" 2>/dev/null | tee -a $OUT/gb_nt_ab.txt
for k in 1 2 3 7; do
  cp build/variants/libarrow_amd_gbnt$k.so arrow_amd/libarrow_amd.so
  echo "== kGbNt=$k" | tee -a $OUT/gb_nt_ab.txt
  timeout 300 python scripts/exp_knobs.py groupby "" 2>/dev/null | tee -a $OUT/gb_nt_ab.txt
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
echo "== tree again" | tee -a $OUT/gb_nt_ab.txt
timeout 300 python scripts/exp_knobs.py groupby "" 2>/dev/null | tee -a $OUT/gb_nt_ab.txt
