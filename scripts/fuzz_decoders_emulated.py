"""Corrupt-input fuzz of the decoders that parse untrusted bytes, on the SIMT emulator (GPU-less): Parquet column chunks of four
writer variants and of the delta encodings through arrow_amd.parquet.read_table, and LZ4-compressed IPC bodies through
arrow_amd.ipc.read_table, each with 1 - 2 flipped bytes inside the data region.  The emulated device buffers are host memory, so
an out-of-bounds lane corrupts the heap or faults: the run must end with "no crash" on every line (a rejected file is fine).
Round 4 found one bug this way (a negative DELTA_BYTE_ARRAY prefix length; tests/test_parquet.py keeps its fuzz case).
    python scripts/fuzz_decoders_emulated.py [seed = 1] [files per variant = 120]"""
import os
import sys
import tempfile

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd  # noqa: E402
from arrow_amd import _lib, array  # noqa: E402
from tests.emu.build_emu import build  # noqa: E402

_lib._lib = _lib.load(build())
array.set_default_device("cpu")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
files = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
n = 4000
words = sorted(("k%05d" % i) * (i % 4) for i in range(n))
t = pa.table({"i": pa.array(rng.integers(0, 50, n), mask=rng.random(n) < 0.2), "f": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1),
              "s": pa.array(words, pa.string(), mask=rng.random(n) < 0.1), "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1)})
VARIANTS = {
    "dictionary + snappy, V2 pages": dict(compression="snappy", data_page_version="2.0", use_dictionary=True, data_page_size=2048),
    "PLAIN + snappy, V1 pages": dict(compression="snappy", data_page_version="1.0", use_dictionary=False, data_page_size=2048),
    "partly dictionary, uncompressed V2": dict(compression="none", data_page_version="2.0", use_dictionary=["s", "i"], data_page_size=1024),
    "DELTA_BINARY_PACKED / BYTE_STREAM_SPLIT / DELTA_LENGTH_BYTE_ARRAY / RLE": dict(
        compression="none", use_dictionary=False, data_page_size=4096,
        column_encoding={"i": "DELTA_BINARY_PACKED", "f": "BYTE_STREAM_SPLIT", "s": "DELTA_LENGTH_BYTE_ARRAY", "b": "RLE"}),
    "DELTA_BYTE_ARRAY": dict(compression="none", use_dictionary=False, data_page_size=2048, column_encoding={"s": "DELTA_BYTE_ARRAY"}),
}
d = tempfile.mkdtemp()
for name, kw in VARIANTS.items():
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
            for v in arrow_amd.parquet.read_table(p2, columns=[t.schema.names[ci]]).values():
                v[0].to_pyarrow()
            ok += 1
        except Exception:
            bad += 1
    print(f"parquet, {name}: no crash; {ok} decoded, {bad} rejected", flush=True)
path = os.path.join(d, "t.arrow")
with ipc.new_file(path, t.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
    for batch in t.to_batches(max_chunksize=1024):
        w.write_batch(batch)
raw = bytearray(open(path, "rb").read())
ok = bad = 0
for it in range(files):
    b = bytearray(raw)
    for _ in range(int(rng.integers(1, 3))):
        b[int(rng.integers(200, len(b) - 400))] = int(rng.integers(0, 256))
    p2 = os.path.join(d, "g.arrow")
    open(p2, "wb").write(b)
    try:
        arrow_amd.ipc.read_table(p2, device_decompress=True)
        ok += 1
    except Exception:
        bad += 1
print(f"ipc, LZ4_FRAME bodies on the device: no crash; {ok} decoded, {bad} rejected", flush=True)
