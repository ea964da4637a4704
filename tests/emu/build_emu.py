"""Builds tests/emu/_build/libarrow_amd_emu.so: the kernel sources of arrow_amd/csrc compiled
for the HOST against the SIMT emulation shim (hip_emu.h).  TEST INFRASTRUCTURE ONLY."""
import contextlib
import fcntl
import glob
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libarrow_amd_emu.so")


@contextlib.contextmanager
def build_lock():
    """One builder at a time (pytest-xdist workers all ask for the library at start-up)."""
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build(force: bool = False) -> str:
    with build_lock():
        return _build(force)


def _build(force: bool) -> str:
    srcs = sorted(glob.glob(os.path.join(ROOT, "arrow_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "arrow_amd", "csrc", "*.h")) + \
        glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.cpp")) + \
        [os.path.join(ROOT, "include", "arrow_amd.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    started = time.time()
    objs, procs = [], []
    for s in srcs + [os.path.join(HERE, "hip_emu_runtime.cpp")]:
        o = os.path.join(HERE, "_build", os.path.basename(s) + ".o")
        cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-fno-strict-aliasing", "-w",
               "-I", HERE, "-x", "c++", "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp = OUT + f".tmp{os.getpid()}"
    subprocess.check_call(["g++", "-shared", "-o", tmp] + objs)
    os.utime(tmp, (started, started))   # a source edited WHILE this build ran must look newer than its output
    os.replace(tmp, OUT)  # a reader never sees a half-written library
    return OUT


if __name__ == "__main__":
    print(build(force=True))
