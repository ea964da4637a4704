"""The headline Filter+Take step through pyarrow.compute on device-resident arrays, a few times, for a kernel-trace
timeline (scripts/rocprof_timeline.py): where the wall time of the CallFunction route goes between the kernels."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pyarrow as pa, pyarrow.compute as pc
from arrow_amd.plugin_build import build_plugin
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
device = torch.device("cuda", 0)
values, validity, mask, _ = bench.gen_filter_inputs(rows, device, 0, 0.10, 0.10)
lib = ctypes.CDLL(build_plugin(verbose=False))
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(k) for k in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
n = rows - rows % 64
hv = pa.Array.from_buffers(pa.int64(), n, [pa.py_buffer(validity[: n // 8].cpu().numpy()), pa.py_buffer(values[: n * 8].cpu().numpy())], null_count=-1)
hm = pa.Array.from_buffers(pa.bool_(), n, [None, pa.py_buffer(mask[: n // 8].cpu().numpy())], null_count=0)
dv, dm = to_device(hv), to_device(hm)
del hv, hm
def step():
    out = pc.filter(dv, dm)
    idx = pc.indices_nonzero(dm)
    return out, pc.take(dv, idx, boundscheck=False)
for i in range(6):
    t0 = time.perf_counter(); r = step(); t1 = time.perf_counter()
    a = time.perf_counter(); o = pc.filter(dv, dm); b = time.perf_counter(); ix = pc.indices_nonzero(dm); c = time.perf_counter(); tk = pc.take(dv, ix, boundscheck=False); d = time.perf_counter()
    print(f"step {i}: {1e3 * (t1 - t0):.3f} ms; parts filter {1e3 * (b - a):.3f} indices_nonzero {1e3 * (c - b):.3f} take {1e3 * (d - c):.3f}", flush=True)
    del r, o, ix, tk
