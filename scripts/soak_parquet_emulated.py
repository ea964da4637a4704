"""Random-shape soak of the plugin's Parquet reader on the EMULATED tier (no GPU): random sizes, null rates, page versions / sizes,
dictionary settings, codecs and row groups; every column through arrow_amd_parquet_read_column and arrow_amd_parquet_read_columns
against pyarrow.  Usage: soak_parquet_emulated.py <first seed> <trials>.  (Found the ring-distance copy of the Snappy LDS decoder.)"""
import ctypes, os, sys, tempfile, faulthandler
faulthandler.enable()
sys.path.insert(0, "/root/repo")
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from tests.emu.build_plugin_emu import build_plugin
lib = ctypes.CDLL(build_plugin(verbose=False))
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
lib.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.arrow_amd_parquet_read_columns.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
assert lib.arrow_amd_register() == 0
def to_host(darr):
    c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(k) for k in (128, 72, 80, 72))
    darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))
def read_column(path, rg, col):
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    rc = lib.arrow_amd_parquet_read_column(path.encode(), rg, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert rc == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    rng = np.random.default_rng(seed0 + trial)
    n = int(rng.integers(1, 30_000))
    null_p = float(rng.choice([0.0, 0.0, 0.1, 0.5, 0.95, 1.0]))
    def m():
        return (rng.random(n) < null_p) if null_p else None
    cols = {"a": pa.array(rng.integers(-2**62, 2**62, n), mask=m()),
            "b": pa.array(rng.integers(0, int(rng.choice([2, 50, 5000])), n).astype(np.int32), mask=m()),
            "c": pa.array(np.round(rng.standard_normal(n), int(rng.integers(0, 4))), mask=m()),
            "d": pa.array(np.cumsum(rng.integers(-3, 4, n)), mask=m()),
            "e": pa.array(rng.standard_normal(n).astype(np.float32)),
            "r": pa.array(rng.integers(0, 9, n))}
    t = pa.table(cols)
    t = t.cast(pa.schema([t.schema.field(i) if t.schema.names[i] not in ("e", "r") else pa.field(t.schema.names[i], t.schema.field(i).type, nullable=False) for i in range(len(t.schema))]))
    opts = dict(compression=str(rng.choice(["snappy", "snappy", "none"])), data_page_version=str(rng.choice(["1.0", "2.0"])),
                use_dictionary=bool(rng.random() < 0.5) if rng.random() < 0.6 else [str(x) for x in rng.choice(list(cols), 2, replace=False)],
                data_page_size=int(rng.choice([512, 4096, 65536, 1 << 20])), dictionary_pagesize_limit=int(rng.choice([1024, 1 << 20])),
                row_group_size=int(rng.choice([n, max(1, n // 2 + 3)])))
    path = os.path.join(tempfile.mkdtemp(), "t.parquet")
    pq.write_table(t, path, **opts)
    pf = pq.ParquetFile(path)
    lib.arrow_amd_plugin_set_parquet_read_threads(int(rng.integers(1, 4)), ctypes.c_int64(4096))
    for rg in range(pf.metadata.num_row_groups):
        ref = pf.read_row_group(rg)
        ncol = len(cols)
        c_devs, c_schemas = ctypes.create_string_buffer(128 * ncol), ctypes.create_string_buffer(72 * ncol)
        rc = lib.arrow_amd_parquet_read_columns(path.encode(), rg, (ctypes.c_int * ncol)(*range(ncol)), ncol, ctypes.addressof(c_devs), ctypes.addressof(c_schemas))
        assert rc == 0, (lib.arrow_amd_plugin_last_error(), opts)
        for ci, name in enumerate(cols):
            w = ref.column(name).combine_chunks()
            h = to_host(read_column(path, rg, ci))
            assert h.equals(w) and h.null_count == w.null_count, (trial, opts, n, null_p, rg, name)
            h2 = to_host(pa.Array._import_from_c_device(ctypes.addressof(c_devs) + 128 * ci, ctypes.addressof(c_schemas) + 72 * ci))
            assert h2.equals(w), (trial, opts, rg, name, "read_columns")
    os.remove(path)
print("PQ_SOAK_OK")
