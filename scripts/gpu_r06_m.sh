#!/bin/bash
# Round 6 call M: is the plugin rebuilt on the GPU box?  Then the whole -m gpu suite again (durations) after the checker work.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_m}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( time python -c "
from arrow_amd.plugin_build import build_plugin
import os, time
from arrow_amd import plugin_build as pb
print('plugin so mtime', os.path.getmtime(pb.OUT), 'src mtime', os.path.getmtime(pb.SRC))
print(build_plugin())" ) > $OUT/plugin_build.txt 2>&1; tail -8 $OUT/plugin_build.txt
( time python -c "
import ctypes
ctypes.CDLL('/opt/rocm/lib/librccl.so')" ) > $OUT/rccl_load.txt 2>&1; tail -4 $OUT/rccl_load.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=40 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -52 $OUT/pytest_gpu.log
