"""Random-shape soak of arrow_amd.ipc.read_table with LZ4_FRAME bodies forced onto the device route, on the EMULATED tier (no GPU):
random sizes (incl. 0), null rates, batch sizes, file / stream format, column subsets, against pyarrow's reader.
Usage: soak_ipc_emulated.py <first seed> <trials>."""
import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np, pyarrow as pa
import arrow_amd
from arrow_amd import _lib, array
from tests.emu.build_emu import build as build_emu
_lib._lib = _lib.load(build_emu())
array.set_default_device("cpu")
seed0 = int(sys.argv[1]); trials = int(sys.argv[2])
words = np.array(["", "a", "bb", "gfx950", "MI355X", "ünïcödé", "x" * 40, "y" * 300], dtype=object)
for trial in range(trials):
    rng = np.random.default_rng(seed0 + trial)
    n = int(rng.integers(0, 40_000))
    null_p = float(rng.choice([0.0, 0.1, 0.6, 1.0]))
    def m():
        return (rng.random(n) < null_p) if null_p else None
    t = pa.table({"i64": pa.array(rng.integers(-2**62, 2**62, n), mask=m()),
                  "small": pa.array(rng.integers(0, 5, n).astype(np.int32), mask=m()),
                  "runs": pa.array(np.repeat(rng.integers(0, 9, n // 50 + 1), 50)[:n]),
                  "f": pa.array(np.round(rng.standard_normal(n), 1), mask=m()),
                  "flag": pa.array(rng.random(n) < 0.3, type=pa.bool_(), mask=m()),
                  "s": pa.array(words[rng.integers(0, len(words), n)], type=pa.string(), mask=m()),
                  "ts": pa.array(rng.integers(0, 2**50, n), pa.timestamp("us"), mask=m())})
    chunk = int(rng.choice([max(1, n // 3 + 1), 1000, 65536, max(1, n)]))
    sink = pa.BufferOutputStream()
    file_format = rng.random() < 0.5
    opts = pa.ipc.IpcWriteOptions(compression="lz4")
    with (pa.ipc.new_file if file_format else pa.ipc.new_stream)(sink, t.schema, options=opts) as w:
        w.write_table(t, max_chunksize=chunk)
    cols = None if rng.random() < 0.5 else [str(x) for x in rng.choice(t.schema.names, 3, replace=False)]
    stats = {}
    got = arrow_amd.ipc.read_table(pa.BufferReader(sink.getvalue()), columns=cols, stats=stats, device_decompress=True)
    ref = (pa.ipc.open_file if file_format else pa.ipc.open_stream)(pa.BufferReader(sink.getvalue())).read_all()
    nb = len(ref.column(0).chunks) if n else 0
    if n:
        assert stats["device_lz4_batches"] == len(got[list(got)[0]]), (stats, n, chunk)
    for name in (cols or t.schema.names):
        chunks = got.get(name, [])
        assert sum(a.length for a in chunks) == n, (name, n)
        whole = pa.chunked_array([a.to_pyarrow() for a in chunks], type=t.schema.field(name).type) if chunks else pa.chunked_array([], type=t.schema.field(name).type)
        assert whole.equals(ref.column(name)), (trial, name, n, chunk, null_p, file_format)
print("IPC_SOAK_OK")
