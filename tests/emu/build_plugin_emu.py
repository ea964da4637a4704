"""TEST BUILD of the Arrow registration shim (arrow_amd/csrc/arrow_plugin.cc + plugin/*.inc) against the
host-emulated kernel library and a host-memory stand-in for the HIP runtime (tests/emu/plugin_hip), so
that the shim's device-resident paths can be exercised without a GPU.  TEST INFRASTRUCTURE ONLY — the
product build is arrow_amd/plugin_build.py."""
import os
import subprocess
import time

from .build_emu import build as build_emu, build_lock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libarrow_amd_plugin_emu.so")


def build_plugin(force: bool = False, verbose: bool = True) -> str:
    core = build_emu()
    with build_lock():
        return _build_plugin(core, force, verbose)


def _build_plugin(core: str, force: bool, verbose: bool) -> str:
    import pyarrow as pa


    src = os.path.join(ROOT, "arrow_amd", "csrc", "arrow_plugin.cc")
    parts = os.path.join(ROOT, "arrow_amd", "csrc", "plugin")
    d = os.path.dirname(pa.__file__)
    so = {name: None for name in ("arrow", "arrow_compute", "arrow_acero", "parquet")}
    for f in sorted(os.listdir(d)):
        for name in so:
            if f.startswith(f"lib{name}.so.") and f.count(".") == 2:
                so[name] = os.path.join(d, f)
    deps = [src, core, os.path.join(ROOT, "include", "arrow_amd.h"),
            os.path.join(HERE, "plugin_hip", "hip", "hip_runtime_api.h")]
    deps += [os.path.join(parts, f) for f in sorted(os.listdir(parts)) if f.endswith(".inc")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(x) for x in deps):
        return OUT
    tmp = OUT + f".tmp{os.getpid()}"
    started = time.time()
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-I", os.path.join(HERE, "plugin_hip"),
           "-I", pa.get_include(), src, "-o", tmp, so["arrow"], so["arrow_compute"], so["arrow_acero"], so["parquet"],
           core, f"-Wl,-rpath,{d}", f"-Wl,-rpath,{os.path.dirname(core)}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.utime(tmp, (started, started))   # a source edited WHILE this (two-minute) build ran must look newer than its output
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build_plugin(force=True))
