#!/bin/bash
# Round 3: result columns copied by a kernel (device-to-device and into mapped pinned host memory) vs the copy engines;
# the gather-form filter with 1 / 2 / 3 / 4 steps in flight, interleaved three times (box drift is ~3 %).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r03_q}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -x -q -m gpu -k "buffer_copy or table_source_rocm or acero" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
ARROW_AMD_AGGREGATE_TIMING=1 timeout 900 python scripts/exp_callfunction_leg.py 300000000 > $OUT/callfunction.json 2> $OUT/callfunction_err.txt; echo "callfunction rc=$?"; grep -A3 "acero\|fused" $OUT/callfunction.json | grep -v "^--"; grep "results to host" $OUT/callfunction_err.txt | sed -n '2p;6p;14p;22p'
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
for rep in 1 2 3; do
  for v in tree u2_w1024 u1_w1024 u3_w1024; do
    if [ $v = tree ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_sparse_$v.so arrow_amd/libarrow_amd.so; fi
    python scripts/exp_streams.py --tag "$v rep$rep" --no-other 2>/dev/null | tee -a $OUT/sparse_ab.jsonl | cut -c1-200
  done
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
