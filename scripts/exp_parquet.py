"""Parquet -> HBM decode of one file vs pyarrow's reader: total time and the device part alone."""
import os, sys, tempfile, time
import numpy as np, torch
import pyarrow as pa, pyarrow.parquet as pq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import parquet as P
n = int(os.environ.get("N", 20_000_000))
rng = np.random.default_rng(1)
t = pa.table({"k": pa.array(rng.integers(0, 5000, n), mask=rng.random(n) < 0.1),
              "v": pa.array(rng.integers(-2**62, 2**62, n)),
              "w": pa.array(np.cumsum(rng.integers(-3, 4, n)))})       # compressible PLAIN int64: the device Snappy route
t = t.cast(pa.schema([pa.field("k", pa.int64()), pa.field("v", pa.int64()), pa.field("w", pa.int64(), nullable=False)]))
path = os.path.join(tempfile.mkdtemp(), "t.parquet")
pq.write_table(t, path, row_group_size=n, compression="snappy", use_dictionary=["k"],
               data_page_version=os.environ.get("PAGE_VERSION", "1.0"))
print("file MB", os.path.getsize(path) / 1e6, flush=True)
t0 = time.perf_counter(); ref = pq.read_table(path, use_threads=False); t_ref = time.perf_counter() - t0
t0 = time.perf_counter(); ref = pq.read_table(path, use_threads=True); t_ref_mt = time.perf_counter() - t0
amd.parquet.read_table(path); torch.cuda.synchronize()
stats = {}
t0 = time.perf_counter(); got = amd.parquet.read_table(path, stats=stats); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("stats", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()})
P.DEVICE_SNAPPY = False
amd.parquet.read_table(path); torch.cuda.synchronize()
t0 = time.perf_counter(); amd.parquet.read_table(path); torch.cuda.synchronize(); t_host_codec = time.perf_counter() - t0
P.DEVICE_SNAPPY = True
print(f"arrow_amd.parquet.read_table with the host Snappy codec for every page: {t_host_codec*1e3:.0f} ms")
for name in ("k", "v", "w"):
    assert got[name][0].to_pyarrow().equals(ref.column(name).combine_chunks()), name
print(f"pyarrow read_table: {t_ref*1e3:.0f} ms (1 thread), {t_ref_mt*1e3:.0f} ms (threads) | arrow_amd.parquet.read_table: {t_all*1e3:.0f} ms "
      f"(host: metadata, page headers, snappy, run-header walk in Python; device: levels, indices, dictionary gather, expand)")
# ---- the C++ binding (arrow_amd_parquet_read_column on parquet::PageReader, plugin/parquet.inc): the same decode
# without the interpreter; one call per column chunk, result = a device-resident pyarrow array
if os.environ.get("CPP_BINDING", "1") == "1":
    import ctypes
    from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin(verbose=False))
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def read_column(col):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        rc = lib.arrow_amd_parquet_read_column(path.encode(), 0, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(k) for k in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    def time_binding(label):
        cols = [read_column(j) for j in range(3)]           # warm-up (pool, page cache, staging block)
        best = 1e9
        per_col = [1e9] * 3
        for _ in range(5):
            t0 = time.perf_counter()
            for j in range(3):
                t1 = time.perf_counter()
                cols[j] = read_column(j)
                per_col[j] = min(per_col[j], time.perf_counter() - t1)
            best = min(best, time.perf_counter() - t0)
        for j, name in enumerate(("k", "v", "w")):
            assert to_host(cols[j]).equals(ref.column(name).combine_chunks()), name
        print(f"arrow_amd_parquet_read_column (C++ on parquet::PageReader; {label}), 3 columns: {best*1e3:.0f} ms "
              f"(k dictionary+nulls {per_col[0]*1e3:.0f}, v PLAIN {per_col[1]*1e3:.0f}, w PLAIN {per_col[2]*1e3:.0f} ms)", flush=True)

    if os.environ.get("ARROW_AMD_PARQUET_TRACE"):
        for j in range(3):
            read_column(j)
            read_column(j)
        sys.exit(0)
    lib.arrow_amd_parquet_read_columns.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    def read_all_columns():
        c_devs, c_schemas = ctypes.create_string_buffer(128 * 3), ctypes.create_string_buffer(72 * 3)
        rc = lib.arrow_amd_parquet_read_columns(path.encode(), 0, (ctypes.c_int * 3)(0, 1, 2), 3, ctypes.addressof(c_devs), ctypes.addressof(c_schemas))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return [pa.Array._import_from_c_device(ctypes.addressof(c_devs) + 128 * i, ctypes.addressof(c_schemas) + 72 * i) for i in range(3)]

    if not os.environ.get("ARROW_AMD_PARQUET_TRACE"):
        for _ in range(3):
            cols = read_all_columns()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter(); cols = read_all_columns(); best = min(best, time.perf_counter() - t0)
        for j, name in enumerate(("k", "v", "w")):
            assert to_host(cols[j]).equals(ref.column(name).combine_chunks()), name
        print(f"arrow_amd_parquet_read_columns (the three chunks at once on the plugin's worker threads): {best*1e3:.0f} ms", flush=True)
        del cols
    time_binding("chunk read once (4 threads) into page-locked memory, Snappy pages decompressed on the device in place")
    lib.arrow_amd_plugin_set_parquet_read_threads(1, ctypes.c_int64(1 << 23))
    time_binding("the same, the chunk read by one thread")
    lib.arrow_amd_plugin_set_parquet_read_threads(8, ctypes.c_int64(1 << 23))
    time_binding("the same, the chunk read by up to 8 threads")
    lib.arrow_amd_plugin_set_parquet_read_threads(4, ctypes.c_int64(1 << 23))
    lib.arrow_amd_plugin_set_parquet_pinned_staging(0)
    time_binding("4 threads, through pageable host memory")
    lib.arrow_amd_plugin_set_parquet_pinned_staging(1)
    lib.arrow_amd_plugin_set_parquet_device_snappy(0)
    time_binding("host codec for every page")
    lib.arrow_amd_plugin_set_parquet_device_snappy(1)
from arrow_amd import tracing
