"""CPU-only check of the Arrow registration shim (arrow_amd/csrc/arrow_plugin.cc): with
ARROW_AMD_PLUGIN_DRY_RUN=1 the kernels are added to Arrow's live registry without a device.
Inputs below the staging threshold are handed to Arrow's own stock kernels (results unchanged);
a call that is routed to the HIP path must fail loudly — there is no CPU compute in the shim."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    from arrow_amd.plugin_build import build_plugin
    so = build_plugin()
    a = pa.array(np.arange(1000), mask=np.arange(1000) % 7 == 0)
    m = pa.array(np.arange(1000) % 3 == 0)
    f = pa.array(np.linspace(-1e40, 1e40, 1000))
    def run():
        return [pc.filter(a, m), pc.take(a, pa.array([5, 1, 999])), pc.greater(f, pa.array(f.to_numpy()[::-1].copy())),
                pc.array_sort_indices(pa.array(np.arange(1000)[::-1].astype(np.uint64))),
                pc.cast(f, pa.float32(), safe=False), pc.cast(f.slice(3), pa.int64(), safe=False)]
    before = pc.get_function("array_filter").num_kernels
    stock = run()
    os.environ["ARROW_AMD_PLUGIN_DRY_RUN"] = "1"
    lib = ctypes.CDLL(so)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    assert lib.arrow_amd_register() == 0            # idempotent
    assert pc.get_function("array_filter").num_kernels > before
    ours = run()
    for x, y in zip(stock, ours):
        assert x.equals(y)
    for fn in (b"array_filter", b"array_take", b"greater", b"array_sort_indices", b"cast"):
        assert lib.arrow_amd_plugin_calls(fn, 0) >= 1, fn     # handed to Arrow's stock kernel
        assert lib.arrow_amd_plugin_calls(fn, 1) == 0, fn     # nothing claimed to be a GPU call
    assert lib.arrow_amd_plugin_calls(b"no_such_function", 0) == -1
    import torch
    if not torch.cuda.is_available():
        # routed to the HIP path without a device: must raise, not compute somewhere else
        lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(10))
        for call in (lambda: pc.filter(a, m), lambda: pc.cast(f, pa.float32(), safe=False),
                     lambda: pa.table({"k": pa.array([1, 2, 1], pa.int32()), "v": pa.array([1, 2, 3], pa.int64())})
                     .group_by("k").aggregate([("v", "sum")])):
            try:
                call()
            except (OSError, pa.ArrowException) as e:
                assert "HIP" in str(e) or "hip" in str(e), str(e)
            else:
                raise SystemExit("a HIP-path call succeeded without a GPU")
    print("REGISTRATION_OK")
''')


def test_registration_dry_run():
    pytest.importorskip("pyarrow")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + SCRIPT], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "REGISTRATION_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
