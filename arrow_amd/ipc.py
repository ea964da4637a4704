"""Arrow IPC files / streams into HBM.

An IPC record batch body already IS the Arrow columnar layout (cpp/src/arrow/ipc/reader.cc:
LoadRecordBatchSubset just points ArrayData at slices of the body), so there is nothing to decode:
the file is memory-mapped, pyarrow's reader resolves the flatbuffer metadata, and every buffer of the
supported column types (fixed width, boolean, utf8 / binary) is copied to the device once —
validity bitmaps, offsets and slice offsets preserved.  (Compressed bodies — LZ4 / ZSTD buffer
compression — are decompressed by pyarrow's reader on the host first.)"""
from __future__ import annotations

from .array import Array


def _batches(source):
    import pyarrow as pa

    if isinstance(source, (str, bytes)) and not isinstance(source, bytes):
        source = pa.memory_map(source, "r")
    try:
        reader = pa.ipc.open_file(source)
        return [reader.get_batch(i) for i in range(reader.num_record_batches)]
    except pa.lib.ArrowInvalid:
        if hasattr(source, "seek"):
            source.seek(0)
        return list(pa.ipc.open_stream(source))


def read_table(source, columns=None, device=None) -> dict:
    """{column name: [device Array per record batch]} of an IPC file (or stream) path / buffer."""
    out: dict = {}
    for batch in _batches(source):
        for name in (batch.schema.names if columns is None else columns):
            out.setdefault(name, []).append(Array.from_pyarrow(batch.column(name), device))
    return out
