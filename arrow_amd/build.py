"""Builds arrow_amd/libarrow_amd.so from arrow_amd/csrc/*.hip with hipcc for gfx950.

`python -m arrow_amd.build [--force]`.  hipcc cross-compiles without a GPU, so this runs in the
CPU-only build container; the .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libarrow_amd.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libarrow_amd.so cannot be built (and there is no fallback)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "arrow_amd.h")]
    if not srcs:
        raise RuntimeError("no kernel sources under arrow_amd/csrc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
