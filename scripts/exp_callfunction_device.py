"""GPU experiment ("also report end-to-end time through CallFunction"): unmodified
pyarrow.compute calls on DEVICE-RESIDENT pyarrow arrays (buffers in HBM, arrow_amd plugin's kROCM
memory manager).  Wall time per call includes Arrow's dispatch, output allocation and the
host-visible count sync."""
import ctypes, sys, os, time, json
import numpy as np
import pyarrow as pa, pyarrow.compute as pc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin
lib = ctypes.CDLL(build_plugin())
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
rng = np.random.default_rng(3)
t0 = time.time()
vals = pa.array(rng.integers(-2**62, 2**62, n, dtype=np.int64), mask=rng.random(n, dtype=np.float32) < 0.1)
mask = pa.array(rng.random(n, dtype=np.float32) < 0.1)
print(f"host arrays in {time.time()-t0:.1f}s", flush=True)
t0 = time.time(); dv, dm = to_device(vals), to_device(mask); print(f"H2D in {time.time()-t0:.2f}s", flush=True)
out = {}
def timeit(name, fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t); del r
    out[name] = {"ms_min": round(min(ts) * 1e3, 3), "ms_median": round(sorted(ts)[len(ts)//2] * 1e3, 3)}
    print(name, out[name], flush=True)
timeit("pc.filter(device int64[%d], device mask 10%%)" % n, lambda: pc.filter(dv, dm))
idx = pc.filter(to_device(pa.array(np.arange(n, dtype=np.uint32))), dm)
timeit("pc.take(device int64, device uint32 indices, no boundscheck)", lambda: pc.take(dv, idx, boundscheck=False))
t = time.perf_counter(); h = pc.filter(vals.slice(0, n // 8), mask.slice(0, n // 8)); cpu_s = time.perf_counter() - t
out["stock CPU pc.filter on the first n/8 host rows (1 thread)"] = {"ms": round(cpu_s * 1e3, 1), "mrows_per_s": round(n / 8 / cpu_s / 1e6, 1)}
print(json.dumps(out))
