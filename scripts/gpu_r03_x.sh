#!/bin/bash
# Round 3: the wide form's aggregate with four records per thread in three 16-byte loads against one record per load,
# interleaved; the probe slice at 2^24 / 2^25 rows instead of 2^26.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_x}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groupby_wide or groupby_probe" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.txt
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
export DEFAULTS="groupby_probe_rows=67108864"
for rep in 1 2; do
  for v in wide narrow; do
    if [ $v = wide ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_gbnarrow.so arrow_amd/libarrow_amd.so; fi
    echo "== $v rep $rep" | tee -a $OUT/gb_ab.txt
    timeout 300 python scripts/exp_knobs.py groupby "" "groupby_probe_rows=16777216" "groupby_probe_rows=33554432" 2>/dev/null | tee -a $OUT/gb_ab.txt
  done
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
