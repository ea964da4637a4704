"""take / filter of a device-resident fixed_size_list<float32>[D] column (embeddings) through unmodified pyarrow.compute
against the reference's kernels on the host copy.  Usage: exp_fsl_take.py [rows = 2^22] [D = 128]"""
import ctypes
import os
import sys
import time

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
emb = pa.FixedSizeListArray.from_arrays(pa.array(rng.standard_normal(n * D, dtype=np.float32)), D)
idx = pa.array(rng.integers(0, n, n // 4).astype(np.uint32))
mask = pa.array(rng.random(n) < 0.25)


def best(fn, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3, out


cpu_take, w_take = best(lambda: pc.take(emb, idx))
cpu_filter, w_filter = best(lambda: pc.filter(emb, mask))
lib = ctypes.CDLL(build_plugin())
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()


def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)


def to_host(x):
    c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
    x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)


d, di, dm = to_device(emb), to_device(idx), to_device(mask)
gpu_take, g_take = best(lambda: pc.take(d, di), 5)
gpu_filter, g_filter = best(lambda: pc.filter(d, dm), 5)
assert to_host(g_take).equals(w_take) and to_host(g_filter).equals(w_filter)
moved = 2 * len(idx) * D * 4
print(f"fixed_size_list<float>[{D}] x {n} rows ({n * D * 4 / 1e9:.2f} GB)")
print(f"take of {len(idx)} random rows : device {gpu_take:8.2f} ms ({moved / gpu_take / 1e6:7.0f} GB/s read + written)   reference on the host {cpu_take:8.1f} ms")
moved = 2 * len(g_filter) * D * 4
print(f"filter, {len(g_filter)} rows kept  : device {gpu_filter:8.2f} ms ({moved / gpu_filter / 1e6:7.0f} GB/s read + written)   reference on the host {cpu_filter:8.1f} ms")
