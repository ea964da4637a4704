#!/bin/bash
# The whole CPU tier in one command: build check + every test that does not need a GPU (oracle pin, C-ABI contract,
# registration, emulator kernel tier, Parquet, emulated plugin tier = the same scripts the GPU tier runs, gloo / fake-RCCL
# world-2).  Run it as the LAST action before the last commit of a work session: every GPU script has an emulated twin, so
# a product change that breaks a GPU test shows here first ("What's weak" 1).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOG=gpurun_out/cpu_gate.log
python scripts/lint_names.py > $LOG 2>&1 || { echo "cpu_gate: lint_names FAILED (a name used in a test / script body is bound nowhere)"; tail -20 $LOG; exit 1; }
python -c "import __graft_entry__ as g; g.build()" >> $LOG 2>&1 || { echo "cpu_gate: build() FAILED"; tail -20 $LOG; exit 1; }
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@" >> $LOG 2>&1
rc=$?
tail -5 $LOG
if [ $rc -eq 0 ]; then echo "cpu_gate: GREEN"; else echo "cpu_gate: RED (rc=$rc) -- see $LOG"; fi
exit $rc
