"""DELTA_BYTE_ARRAY column chunks decoded into HBM (arrow_amd.parquet.read_table: one wave per page walks the values)
against pyarrow's reader on the host cores: sorted keys that share long prefixes, by page size.
Usage: exp_parquet_dba.py [rows = 10_000_000]"""
import os
import sys
import tempfile
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402

import arrow_amd as amd  # noqa: E402
from arrow_amd.plugin_build import build_plugin  # noqa: E402

plug = ctypes.CDLL(build_plugin())
plug.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
plug.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def read_column_cpp(path):
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    rc = plug.arrow_amd_parquet_read_column(path.encode(), 0, 0, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert rc == 0, plug.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(1)
ids = np.sort(rng.integers(0, 100 * n, n))
keys = pa.array(np.char.add("tenant-0042/2026-09-24/object/", np.char.zfill(ids.astype(str), 12)), pa.string())
t = pa.table({"k": keys})
print(f"{n} keys of {len(keys[0].as_py())} bytes, {keys.nbytes / 1e6:.0f} MB as an array", flush=True)
for page_size in (1 << 20, 1 << 16):
    path = os.path.join(tempfile.mkdtemp(), "k.parquet")
    pq.write_table(t, path, use_dictionary=False, column_encoding={"k": "DELTA_BYTE_ARRAY"}, compression="snappy",
                   data_page_size=page_size, row_group_size=n)
    size = os.path.getsize(path)
    for threads in (False, True):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ref = pq.read_table(path, use_threads=threads)
            best = min(best, time.perf_counter() - t0)
        print(f"page {page_size >> 10:5d} KB  file {size / 1e6:7.1f} MB  pyarrow use_threads={threads}: {best * 1e3:8.1f} ms", flush=True)
    best, stats = 1e9, {}
    for _ in range(4):
        st = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = amd.parquet.read_table(path, stats=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt < best:
            best, stats = dt, st
    bestc = 1e9
    for _ in range(4):      # (the registration of the plugin comes after pyarrow's own timings: its reader is unaffected anyway)
        if _ == 0:
            assert plug.arrow_amd_register() == 0
        t0 = time.perf_counter()
        dcol = read_column_cpp(path)
        torch.cuda.synchronize()
        bestc = min(bestc, time.perf_counter() - t0)
    assert len(dcol) == n
    print(f"page {page_size >> 10:5d} KB  arrow_amd_parquet_read_column (C++ shim) into HBM: {bestc * 1e3:8.1f} ms", flush=True)
    del dcol
    arr = got["k"][0]
    assert arr.to_pyarrow().equals(ref.column("k").combine_chunks())
    print(f"page {page_size >> 10:5d} KB  arrow_amd.parquet.read_table into HBM: {best * 1e3:8.1f} ms   (host prep {stats.get('host_prep_s', 0) * 1e3:.1f} ms)",
          flush=True)
