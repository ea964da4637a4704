#!/bin/bash
# Round 4: group-by variants A/B at 4e9 rows / 1e7 keys, interleaved: VARIANTS = names under build/variants (tree = the tree's lib)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_g}
mkdir -p $OUT
export TMPDIR=/tmp
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
for rep in 1 2; do
  for v in ${VARIANTS:-tree}; do
    if [ $v = tree ]; then cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so; else cp build/variants/libarrow_amd_$v.so arrow_amd/libarrow_amd.so; fi
    echo "== $v rep $rep" | tee -a $OUT/gb_ab.txt
    ${ENVV:-} timeout 300 python scripts/exp_knobs.py ${WHAT:-groupby} "" 2>/dev/null | tee -a $OUT/gb_ab.txt
  done
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
