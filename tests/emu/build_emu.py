"""Builds tests/emu/_build/libarrow_amd_emu.so: the kernel sources of arrow_amd/csrc compiled
for the HOST against the SIMT emulation shim (hip_emu.h).  TEST INFRASTRUCTURE ONLY."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libarrow_amd_emu.so")


def build(force: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(ROOT, "arrow_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "arrow_amd", "csrc", "*.h")) + \
        glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.cpp")) + \
        [os.path.join(ROOT, "include", "arrow_amd.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    for s in srcs + [os.path.join(HERE, "hip_emu_runtime.cpp")]:
        o = os.path.join(HERE, "_build", os.path.basename(s) + ".o")
        cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-fno-strict-aliasing", "-w",
               "-I", HERE, "-x", "c++", "-c", s, "-o", o]
        subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call(["g++", "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
