"""Corrupt-input fuzz of round 6's decoders on the SIMT emulator (GPU-less; see fuzz_decoders_emulated.py): Parquet list and
struct columns (levels of any width, DefRepLevelsToList, the leaf filter) and GZIP pages on the device route, each with 1 - 2
flipped bytes inside a column chunk.  A rejected file is fine; the run must end with "no crash" on every line.
    python scripts/fuzz_nested_gzip_emulated.py [seed = 1] [files per variant = 60]"""
import os
import sys
import tempfile

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd  # noqa: E402
from arrow_amd import _lib, array  # noqa: E402
from tests.emu.build_emu import build  # noqa: E402
from tests.test_parquet import _list_table, _struct_table  # noqa: E402

_lib._lib = _lib.load(build())
array.set_default_device("cpu")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
files = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
n = 1500
flat = pa.table({"a": pa.array(np.cumsum(rng.integers(-3, 4, n))), "b": pa.array(rng.integers(0, 50, n).astype(np.int32)),
                 "c": pa.array(np.round(rng.standard_normal(n), 1), mask=rng.random(n) < 0.1)})
flat = flat.cast(pa.schema([pa.field("a", pa.int64(), nullable=False), pa.field("b", pa.int32(), nullable=False), pa.field("c", pa.float64())]))
CASES = {
    "lists, V1 pages, snappy": (_list_table(rng, n, 0.15), dict(compression="snappy", data_page_version="1.0", data_page_size=1024)),
    "lists, V2 pages, PLAIN": (_list_table(rng, n, 0.15), dict(compression="none", data_page_version="2.0", use_dictionary=False, data_page_size=1024)),
    "structs, V2 pages": (_struct_table(rng, n, 0.15, 0.2), dict(compression="none", data_page_version="2.0", data_page_size=1024)),
    "GZIP PLAIN pages, V2": (flat, dict(compression="gzip", data_page_version="2.0", use_dictionary=False, data_page_size=2048)),
    "GZIP PLAIN pages, V1": (flat, dict(compression="gzip", data_page_version="1.0", use_dictionary=False, data_page_size=2048)),
}
d = tempfile.mkdtemp()
for name, (t, kw) in CASES.items():
    path = os.path.join(d, "f.parquet")
    pq.write_table(t, path, **kw)
    raw = bytearray(open(path, "rb").read())
    md = pq.ParquetFile(path).metadata
    ok = bad = 0
    for it in range(files):
        ci = it % md.num_columns
        col = md.row_group(0).column(ci)
        lo = col.dictionary_page_offset if col.has_dictionary_page else col.data_page_offset
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(lo, lo + col.total_compressed_size))] = int(rng.integers(0, 256))
        p2 = os.path.join(d, "g.parquet")
        open(p2, "wb").write(b)
        try:
            for v in arrow_amd.parquet.read_table(p2, columns=[md.schema.column(ci).path]).values():
                v[0].to_pyarrow()
            ok += 1
        except Exception:
            bad += 1
    print(f"{name}: {files} corrupted files, {ok} still decoded, {bad} rejected, no crash", flush=True)
