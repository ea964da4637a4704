"""Acero plans over device-resident tables: table_source -> filter -> aggregate_rocm, all in HBM."""
import ctypes, faulthandler, os, sys, time
import numpy as np
import pyarrow as pa, pyarrow.compute as pc
from pyarrow import acero
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin
lib = ctypes.CDLL(build_plugin())
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

rng = np.random.default_rng(5)
n = int(os.environ.get("N", 4_000_000))
k = pa.array(rng.integers(-5000, 5000, n).astype(np.int32), mask=rng.random(n) < 0.01)
v = pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.05)
w = pa.array(rng.integers(-100, 100, n))
host = pa.table({"k": k, "v": v, "w": w})
dev = pa.table({"k": to_device(k), "v": to_device(v), "w": to_device(w)})
print("device table built", flush=True)

def plan(table, agg):
    return acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
        acero.Declaration("filter", acero.FilterNodeOptions(pc.field("w") > 10)),
        acero.Declaration(agg, acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"])),
    ])

want = plan(host, "aggregate").to_table(use_threads=False).sort_by("k")
print("host plan done", flush=True)
g0 = {f: lib.arrow_amd_plugin_calls(f, 1) for f in (b"greater", b"array_filter", b"hash_sum")}
t0 = time.perf_counter()
got = plan(dev, "aggregate_rocm").to_table(use_threads=False)
dt = time.perf_counter() - t0
print("device plan done in %.1f ms" % (dt * 1e3), flush=True)
g1 = {f: lib.arrow_amd_plugin_calls(f, 1) for f in g0}
print({f.decode(): g1[f] - g0[f] for f in g0})
got = got.sort_by("k")
assert got.schema.names == ["k", "v_sum"]
assert got.equals(want.select(["k", "v_sum"])), (got.slice(0, 5), want.slice(0, 5))
# A/B of the single-synchronisation filter path (off by default until it has been measured here)
lib.arrow_amd_plugin_set_filter_morsel_rows.argtypes = [ctypes.c_int64]
for morsel in (0, 1 << 20, 0, 1 << 20):
    lib.arrow_amd_plugin_set_filter_morsel_rows(morsel)
    for threads in (False, True):
        t0 = time.perf_counter()
        again = plan(dev, "aggregate_rocm").to_table(use_threads=threads)
        dt = time.perf_counter() - t0
        assert again.sort_by("k").equals(got)
        print("filter_morsel_rows=%d threads=%s: device plan %.1f ms" % (morsel, threads, dt * 1e3), flush=True)
lib.arrow_amd_plugin_set_filter_morsel_rows(0)
print("ACERO_DEVICE_OK")
