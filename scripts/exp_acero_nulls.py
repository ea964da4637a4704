"""aggregate_rocm over a device-resident table whose columns have nulls: batches staged many at a time (validity by
arx_bitmap_copy_segments) vs every 32K-row batch consumed on its own (the route before), and the null-free plan."""
import ctypes, os, sys, time
import numpy as np, pyarrow as pa, pyarrow.acero as acero, pyarrow.compute as pc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin
lib = ctypes.CDLL(build_plugin(verbose=False))
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()


def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(k) for k in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)


n = int(os.environ.get("N", 1 << 26))
groups = int(os.environ.get("GROUPS", 1_000_000))
rng = np.random.default_rng(2)
keys = rng.integers(0, groups, n).astype(np.int32)
vals = rng.integers(-2**40, 2**40, n)
for null_p in (0.0, 0.1):
    k = pa.array(keys, mask=(rng.random(n) < null_p / 10) if null_p else None)
    v = pa.array(vals, mask=(rng.random(n) < null_p) if null_p else None)
    dev = pa.table({"k": to_device(k), "v": to_device(v)})
    plan = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(dev)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))])
    want = None
    for stage in ((1,) if null_p == 0 else (1, 0)):
        lib.arrow_amd_plugin_set_aggregate_stage_nulls(stage)
        got = plan.to_table(use_threads=False)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); got = plan.to_table(use_threads=False); ts.append(time.perf_counter() - t0)
        got = got.sort_by("k")
        if want is None:
            want = got
        assert got.equals(want)
        print(f"aggregate_rocm, {n} device rows, {groups} keys, {null_p:.0%} null values: "
              f"{'batches staged' if stage else 'every batch with nulls consumed on its own'}: {min(ts)*1e3:8.1f} ms", flush=True)
    lib.arrow_amd_plugin_set_aggregate_stage_nulls(1)
