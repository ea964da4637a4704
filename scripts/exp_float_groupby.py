"""hash_sum / hash_mean of a float64 column by an int32 key through aggregate_rocm over a device-resident table against the
reference's GroupByNode on the host copy (one thread: the order the device reproduces; and all threads).
Usage: exp_float_groupby.py [rows = 2^26] [keys = 1000000]"""
import ctypes
import os
import sys
import time

import numpy as np
import pyarrow as pa
from pyarrow import acero

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rng = np.random.default_rng(0)
t = pa.table({"k": pa.array(rng.integers(0, K, n).astype(np.int32)), "v": pa.array(rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n))})
aggs = [("v", "hash_sum", None, "s"), ("v", "hash_mean", None, "m")]


def plan(tab, node, threads):
    return acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
        acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=["k"]))]).to_table(use_threads=threads)


def best(fn, reps=2):
    b, out = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3, out


cpu1, w = best(lambda: plan(t, "aggregate", False), 1)
cpuN, _ = best(lambda: plan(t, "aggregate", True))
lib = ctypes.CDLL(build_plugin())
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()


def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)


td = pa.table({c: to_device(t.column(c).chunk(0)) for c in ("k", "v")})
gpu, g = best(lambda: plan(td, "aggregate_rocm", False), 3)
w, g = w.sort_by("k"), g.sort_by("k")
same = np.asarray(w.column("s")).view(np.uint64) == np.asarray(g.column("s")).view(np.uint64)
print(f"{n} rows, {K} keys, float64 values: aggregate_rocm over the device table {gpu:8.1f} ms = {n / gpu / 1e6:6.2f} Grows/s;  "
      f"reference GroupByNode 1 thread {cpu1:9.1f} ms, all threads {cpuN:8.1f} ms;  sums bit-identical to the 1-thread reference: {bool(same.all())}")
