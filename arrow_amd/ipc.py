"""Arrow IPC files / streams into HBM.

An IPC record batch body already IS the Arrow columnar layout (cpp/src/arrow/ipc/reader.cc: LoadRecordBatchSubset
points ArrayData at slices of the body), so an uncompressed body only has to be copied.  A body with BUFFER COMPRESSION
(RecordBatch.compression, ipc/reader.cc DecompressBuffers: every buffer = int64 uncompressed length + one LZ4 frame,
or -1 + the raw bytes) is decompressed ON THE DEVICE for LZ4_FRAME: this module reads the messages itself
(`pyarrow.ipc.MessageReader` for the framing, the RecordBatch flatbuffer — format/Message.fbs — parsed below for the
field nodes, the buffer table and the codec), uploads the body once, walks every needed buffer's frame header on the
host (arx_lz4_frame_scan: a few bytes per 64 KB block) and decodes all buffers of a batch in one launch
(arx_lz4_decompress_streams: one wave per buffer).  ZSTD bodies, dictionary batches and nested / large types go through
pyarrow's reader (host decompression) and a copy, as before.

Supported on the device route: fixed-width, boolean, utf8 / binary columns (validity bitmaps and offsets preserved)."""
from __future__ import annotations

import ctypes as C
import struct
import warnings

import numpy as np

from . import _lib
from .array import Array, alloc, current_stream, default_device, is_base_binary, to_device, type_from_name

LZ4_BLOCK = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("stored", "<u4")])
LZ4_STREAM = np.dtype([("first_block", "<u8"), ("num_blocks", "<u4"), ("reserved", "<u4"), ("dst_offset", "<u8"),
                       ("dst_size", "<u8")])
_MESSAGE_RECORD_BATCH = 3          # MessageHeader union (format/Message.fbs): 1 Schema, 2 DictionaryBatch, 3 RecordBatch
_CODEC_LZ4_FRAME = 0               # CompressionType: 0 LZ4_FRAME, 1 ZSTD


# ------------------------------------------------------------------ flatbuffer access (format/Message.fbs)
class _Table:
    """A flatbuffer table: field i lives at table + vtable[i] (0 = absent)."""

    def __init__(self, buf: bytes, pos: int):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        if self.vt < 0:
            raise IndexError("flatbuffer vtable before the metadata")
        self.vt_size = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, i: int) -> int:
        at = 4 + 2 * i
        return struct.unpack_from("<H", self.buf, self.vt + at)[0] if at + 2 <= self.vt_size else 0

    def scalar(self, i: int, fmt: str, default=0):
        o = self._off(i)
        return struct.unpack_from(fmt, self.buf, self.pos + o)[0] if o else default

    def table(self, i: int):
        o = self._off(i)
        if not o:
            return None
        p = self.pos + o
        return _Table(self.buf, p + struct.unpack_from("<I", self.buf, p)[0])

    def struct_vector(self, i: int, fmt: str, width: int):
        o = self._off(i)
        if not o:
            return []
        p = self.pos + o
        v = p + struct.unpack_from("<I", self.buf, p)[0]
        n = struct.unpack_from("<I", self.buf, v)[0]
        if v + 4 + n * width > len(self.buf):
            raise IndexError("flatbuffer vector runs past the metadata")
        return [struct.unpack_from(fmt, self.buf, v + 4 + k * width) for k in range(n)]


def parse_record_batch_message(metadata: bytes):
    """Message flatbuffer -> None (not a record batch) or dict(length, nodes [(length, null_count)], buffers
    [(offset, length)], codec (None | 0 LZ4_FRAME | 1 ZSTD))."""
    msg = _Table(metadata, struct.unpack_from("<I", metadata, 0)[0])
    if msg.scalar(1, "<B") != _MESSAGE_RECORD_BATCH:
        return None
    rb = msg.table(2)
    if rb is None:
        raise IndexError("record batch message without a header")
    comp = rb.table(3)
    return {"length": rb.scalar(0, "<q"), "nodes": rb.struct_vector(1, "<qq", 16), "buffers": rb.struct_vector(2, "<qq", 16),
            "codec": None if comp is None else comp.scalar(0, "<b"), "variadic": bool(rb._off(4))}


def _buffers_per_field(t) -> int | None:
    """Buffers a flat column contributes to the body (ipc/reader.cc ArrayLoader); None = not a type this route handles."""
    import pyarrow as pa

    if pa.types.is_string(t) or pa.types.is_binary(t):
        return 3
    try:
        at = type_from_name(str(t))
    except Exception:
        return None
    return None if is_base_binary(at) else 2


# ------------------------------------------------------------------ the device route
def _plan_batch_lz4(lib, schema, info, body, columns, body_base, dst_base, streams, blocks, raw_copies):
    """The buffers of one record batch with LZ4_FRAME compression: frame headers walked on the host, one stream per
    compressed buffer appended to `streams` / `blocks` (positions relative to the file-wide device copies).  Returns
    (columns [(field, node, [slot per buffer])], bytes of output taken) or None if a column type is not handled."""
    plan, bi = [], 0
    for fi, field in enumerate(schema):
        nb = _buffers_per_field(field.type)
        if nb is None:
            return None
        plan.append((field, info["nodes"][fi], info["buffers"][bi: bi + nb]))
        bi += nb
    if bi != len(info["buffers"]) or info["variadic"]:
        return None
    body_np = np.frombuffer(body, dtype=np.uint8) if body is not None and body.size else np.zeros(0, np.uint8)
    cols, dst = [], dst_base
    for field, node, bufs in plan:
        if columns is not None and field.name not in columns:
            continue
        slots = []
        for off, length in bufs:
            if length == 0:
                slots.append(None)
                continue
            if length < 8 or off < 0 or off + length > len(body_np):
                raise _lib.ArrowInvalid("IPC: a buffer lies outside the message body")
            ulen = struct.unpack_from("<q", body_np, off)[0]     # DecompressBuffers: the uncompressed length prefix
            if ulen == -1:                                       # stored as it is
                ulen = length - 8
                raw_copies.append((dst, body_base + off + 8, ulen))
            else:
                if ulen < 0 or ulen > 255 * length + (1 << 16):   # (LZ4 cannot expand more than 255 : 1)
                    raise _lib.ArrowInvalid("IPC: implausible uncompressed buffer length")
                frame = body_np[off + 8: off + length]
                nblk = C.c_int64(0)
                _lib.check(lib.arx_lz4_frame_scan(frame.ctypes.data, len(frame), body_base + off + 8, None, 0, C.byref(nblk), None))
                tab = np.zeros(max(nblk.value, 1), LZ4_BLOCK)
                _lib.check(lib.arx_lz4_frame_scan(frame.ctypes.data, len(frame), body_base + off + 8, tab.ctypes.data, nblk.value,
                                                  C.byref(nblk), None))
                streams.append((len(blocks), nblk.value, 0, dst, ulen))
                blocks.extend(tab[: nblk.value].tolist())
            slots.append((dst, ulen))
            dst += (ulen + 63) & ~63                              # every buffer on its own 64-byte boundary
        cols.append((field, node, slots))
    return cols, dst - dst_base


def _decode_file_lz4(schema, msgs, infos, columns, device, force: bool):
    """Every record batch of the file in ONE decode launch (one wave per buffer: the parallelism is the number of
    buffers).  Returns None when a column type is not handled, or — unless forced — when the file has too few / too
    large buffers for the device to win (a wave decodes ~16 MB/s; the reference's host codec runs at GB/s per core)."""
    import torch

    lib, stream = _lib.get_lib(), current_stream(device)
    streams, blocks, raw_copies, per_batch = [], [], [], []
    body_base = dst_base = 0
    bodies = []
    for m, info in zip(msgs, infos):
        planned = _plan_batch_lz4(lib, schema, info, m.body, columns, body_base, dst_base, streams, blocks, raw_copies)
        if planned is None:
            return None
        cols, used = planned
        per_batch.append((info, cols))
        n = m.body.size if m.body is not None else 0
        bodies.append((body_base, m.body, n))
        body_base += (n + 63) & ~63
        dst_base += used
    if not force and streams:
        largest = max(s[4] for s in streams)
        if len(streams) < 512 or largest > (1 << 20):
            return None
    d_body = alloc(body_base + 64, device)
    for at, body, n in bodies:
        if n:
            with warnings.catch_warnings():    # (a read-only view of the mapped file; it is only read)
                warnings.simplefilter("ignore", UserWarning)
                d_body[at: at + n].copy_(torch.from_numpy(np.frombuffer(body, dtype=np.uint8)))
    out = alloc(dst_base + 64, device)
    for d, s_, n in raw_copies:
        out[d: d + n] = d_body[s_: s_ + n]
    if streams:
        st_np, bl_np = np.array(streams, LZ4_STREAM), (np.array(blocks, LZ4_BLOCK) if blocks else np.zeros(1, LZ4_BLOCK))
        d_streams, d_blocks = to_device(st_np.view(np.uint8), device), to_device(bl_np.view(np.uint8), device)
        status = alloc(len(streams) * 4, device, zero=True)
        _lib.check(lib.arx_lz4_decompress_streams(d_body.data_ptr(), d_streams.data_ptr(), d_blocks.data_ptr(),
                                                  len(streams), out.data_ptr(), status.data_ptr(), stream))
        if int(status.view(dtype=_i32()).max().item()) != 0:
            raise OSError("LZ4 decompress failed: corrupt IPC buffer")   # Lz4FrameCodec::Decompress, compression_lz4.cc
    decoded = []
    for info, cols in per_batch:
        result = {}
        for field, (length, null_count), slots in cols:
            t = type_from_name(str(field.type))
            views = [None if s_ is None else out[s_[0]: s_[0] + s_[1]] for s_ in slots]
            validity = views[0] if null_count != 0 else None
            if len(slots) == 3:
                offsets = views[1] if views[1] is not None else to_device(np.zeros(1, np.int32), device)
                data = views[2] if views[2] is not None else alloc(0, device)
                result[field.name] = Array(t, length, [validity, offsets, data], null_count if validity is not None else 0, 0)
            else:
                data = views[1] if views[1] is not None else alloc(0, device)
                result[field.name] = Array(t, length, [validity, data], null_count if validity is not None else 0, 0)
        decoded.append(result)
    return decoded


def _i32():
    import torch

    return torch.int32


def _messages(source):
    """(schema, [pyarrow Message of every record batch]) of an IPC file or stream, or None when the messages cannot be
    read one by one (then the whole source goes through pyarrow's reader)."""
    import pyarrow as pa

    # the bytes are read ONCE, outside the try: a file-like source is consumed by read_buffer(), and the fallback must
    # get the same bytes back (it used to re-read an exhausted stream)
    if isinstance(source, str):
        source = pa.memory_map(source, "r")
    buf = source.read_buffer() if hasattr(source, "read_buffer") else pa.py_buffer(source)
    try:
        is_file = buf.size >= 8 and buf.slice(0, 6).to_pybytes() == b"ARROW1"
        reader = pa.ipc.MessageReader.open_stream(pa.BufferReader(buf.slice(8) if is_file else buf))
        schema, batches = None, []
        for m in reader:
            if m.type == "schema":
                schema = pa.ipc.read_schema(pa.BufferReader(m.serialize()))
            elif m.type == "record batch":
                batches.append(m)
            else:
                return None, buf                                  # dictionary batches: pyarrow's reader
        return (schema, batches), buf
    except (pa.ArrowInvalid, OSError):
        return None, buf                                          # the message walk failed: pyarrow's reader, same bytes


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


def read_table(source, columns=None, device=None, device_decompress="auto", stats: dict | None = None) -> dict:
    """{column name: [device Array per record batch]} of an IPC file (or stream) path / buffer.
    device_decompress: True = LZ4_FRAME bodies are always decompressed on the device; "auto" (default) = only when the
    file has many small buffers (>= 512 of <= 1 MB: one wave decodes one buffer, at a fraction of a host core's rate,
    so the device wins through parallelism or not at all); False = pyarrow's reader decompresses on the host.
    stats (optional dict): "device_lz4_batches" = record batches whose buffers were decompressed on the device."""
    import pyarrow as pa

    device = default_device() if device is None else device
    out: dict = {}
    parsed, buf = _messages(source) if device_decompress else (None, None)
    if parsed is not None and parsed[0] is not None:
        schema, msgs = parsed
        if columns is not None:
            missing = [c for c in columns if c not in schema.names]
            if missing:
                raise KeyError(f"read_table: column(s) {missing} not in the file's schema {schema.names}")
        try:
            infos = [parse_record_batch_message(m.metadata.to_pybytes()) for m in msgs]
        except (struct.error, IndexError):      # metadata this parser cannot follow: the reference's reader decides
            infos = []
        if infos and all(i is not None and i["codec"] == _CODEC_LZ4_FRAME for i in infos):
            names = schema.names if columns is None else list(columns)
            decoded = _decode_file_lz4(schema, msgs, infos, None if columns is None else set(columns), device,
                                       force=device_decompress is True)
            if decoded is not None:
                for d in decoded:
                    for name in names:
                        out.setdefault(name, []).append(d[name])
                if stats is not None:
                    stats["device_lz4_batches"] = len(decoded)
                return out
    if buf is not None and not isinstance(source, str):
        source = pa.BufferReader(buf)
    for batch in _batches(source):
        for name in (batch.schema.names if columns is None else columns):
            out.setdefault(name, []).append(Array.from_pyarrow(batch.column(name), device))
    if stats is not None:
        stats["device_lz4_batches"] = 0
    return out
