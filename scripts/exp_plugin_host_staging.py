"""Host-resident pyarrow arrays through the plugin (PCIe staging path) vs Arrow's stock CPU kernels,
same process: arrow_amd_plugin_set_min_rows switches the route."""
import ctypes, os, sys, time
import numpy as np
import pyarrow as pa, pyarrow.compute as pc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin
lib = ctypes.CDLL(build_plugin())
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
rng = np.random.default_rng(1)
n = int(os.environ.get("N", 100_000_000))
vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
mask = pa.array(rng.random(n) < 0.1)
idx = pa.array(np.sort(rng.integers(0, n, n // 10)).astype(np.uint32))
f64 = pa.array(rng.standard_normal(n))
def bench(fn, reps=3):
    fn(); t = time.perf_counter()
    for _ in range(reps): out = fn()
    return (time.perf_counter() - t) / reps, out
for name, fn in (("filter", lambda: pc.filter(vals, mask)), ("take", lambda: pc.take(vals, idx)),
                 ("cast f64->f32", lambda: pc.cast(f64, pa.float32())), ("greater", lambda: pc.greater(f64, f64)),
                 ("sort_indices", lambda: pc.array_sort_indices(vals))):
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(1 << 62))
    t_cpu, ref = bench(fn, 2)
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(1 << 16))
    g0 = lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1) + lib.arrow_amd_plugin_calls(b"cast", 1) + lib.arrow_amd_plugin_calls(b"greater", 1) + lib.arrow_amd_plugin_calls(b"array_sort_indices", 1)
    t_gpu, got = bench(fn, 3)
    g1 = lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1) + lib.arrow_amd_plugin_calls(b"cast", 1) + lib.arrow_amd_plugin_calls(b"greater", 1) + lib.arrow_amd_plugin_calls(b"array_sort_indices", 1)
    assert got.equals(ref), name
    print(f"{name:14s} n={n}: stock CPU {t_cpu*1e3:8.1f} ms | plugin (PCIe staging) {t_gpu*1e3:8.1f} ms | x{t_cpu/t_gpu:.1f} | gpu calls {g1-g0}", flush=True)
