#!/bin/bash
# Round 3: A/B of the selection kernels' cache policy (kSelNt variants of libarrow_amd.so, scripts/build_sel_variants.sh)
# with the one-shot grids of scalar.hip / take in every variant.  The tree's own library (kSelNt = 3) runs first and last.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_sel_nt
mkdir -p $OUT
export TMPDIR=/tmp
cp arrow_amd/libarrow_amd.so /tmp/lib_tree.so
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or take or cast or greater" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
python scripts/exp_streams.py --tag "tree(kSelNt=3)" 2>/dev/null | tee -a $OUT/ab.jsonl
for k in 0 1 7; do
  cp build/variants/libarrow_amd_selnt$k.so arrow_amd/libarrow_amd.so
  python scripts/exp_streams.py --tag "kSelNt=$k" 2>/dev/null | tee -a $OUT/ab.jsonl
done
cp /tmp/lib_tree.so arrow_amd/libarrow_amd.so
python scripts/exp_streams.py --tag "tree(kSelNt=3) again" --no-other 2>/dev/null | tee -a $OUT/ab.jsonl
