"""Device-resident Arrow arrays: the host-side mirror of arrow::ArrayData / ArraySpan
(cpp/src/arrow/array/data.h:85,553) for buffers that live in MI355X HBM.

PyTorch is plumbing here: a `torch.uint8` tensor is the owner of a device allocation
(the role arrow::Buffer + MemoryPool play on the host, cpp/src/arrow/buffer.h,
memory_pool.h); nothing in this module computes with torch.
Layout is Arrow's columnar format (docs/source/format/Columnar.rst): buffers[0] = validity
bitmap (LSB-first, optional), buffers[1] = fixed-width values (a bitmap for boolean),
plus a logical `offset` and `length`.  Allocations are padded to 64 bytes like Arrow's.
"""
from __future__ import annotations

import ctypes as C

import re

import numpy as np
import torch

from . import _lib

kUnknownNullCount = -1

_default_device = None


def set_default_device(device) -> None:
    """Device on which new buffers are allocated (default: current CUDA/HIP device)."""
    global _default_device
    _default_device = None if device is None else torch.device(device)


def default_device() -> torch.device:
    if _default_device is not None:
        return _default_device
    if not torch.cuda.is_available():
        raise _lib.ArrowDeviceError(
            "arrow_amd needs a HIP device (MI355X): no GPU is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def current_stream(device: torch.device) -> int:
    """hipStream_t (as int) the C ABI enqueues on: torch's current stream of `device`."""
    if device.type == "cuda":
        return int(torch.cuda.current_stream(device).cuda_stream)
    return 0


# ------------------------------------------------------------------ types
class DataType:
    """Twin of arrow::DataType for the fixed-width types on the hot path."""

    def __init__(self, name: str, bit_width: int, np_dtype):
        self.name = name
        self.bit_width = bit_width
        self.np_dtype = None if np_dtype is None else np.dtype(np_dtype)

    @property
    def byte_width(self) -> int:
        return self.bit_width // 8

    def __repr__(self):
        return f"DataType({self.name})"

    def __eq__(self, other):
        return isinstance(other, DataType) and other.name == self.name

    def __hash__(self):
        return hash(self.name)


bool_ = DataType("bool", 1, np.bool_)
int8 = DataType("int8", 8, np.int8)
uint8 = DataType("uint8", 8, np.uint8)
int16 = DataType("int16", 16, np.int16)
uint16 = DataType("uint16", 16, np.uint16)
int32 = DataType("int32", 32, np.int32)
uint32 = DataType("uint32", 32, np.uint32)
int64 = DataType("int64", 64, np.int64)
uint64 = DataType("uint64", 64, np.uint64)
float32 = DataType("float", 32, np.float32)
float64 = DataType("double", 64, np.float64)

# base binary types (int32 offsets): buffers = [validity, offsets, data]
binary = DataType("binary", 0, None)
utf8 = DataType("string", 0, None)
string = utf8

_ALL_TYPES = [bool_, int8, uint8, int16, uint16, int32, uint32, int64, uint64, float32, float64]
_BY_NAME = {t.name: t for t in _ALL_TYPES}
_BY_NAME.update({"float32": float32, "float64": float64, "boolean": bool_, "binary": binary,
                 "string": utf8, "utf8": utf8})
_BY_NP = {t.np_dtype: t for t in _ALL_TYPES}


# temporal types: the layout of their physical integer (what filter / take / sort / the Parquet decode move around)
_TEMPORAL_64 = re.compile(r"^(timestamp\[(s|ms|us|ns)(, tz=.+)?\]|duration\[(s|ms|us|ns)\]|time64\[(us|ns)\]|date64\[ms\])$")
_TEMPORAL_32 = re.compile(r"^(date32\[day\]|time32\[(s|ms)\])$")


def is_temporal(t: DataType) -> bool:
    return bool(_TEMPORAL_64.match(t.name) or _TEMPORAL_32.match(t.name))


def is_base_binary(t: DataType) -> bool:
    return t.name in ("binary", "string")

# index type ids of include/arrow_amd.h
INDEX_TYPE_ID = {"uint8": 0, "int8": 1, "uint16": 2, "int16": 3, "uint32": 4, "int32": 5,
                 "uint64": 6, "int64": 7}


def type_from_name(name: str) -> DataType:
    try:
        return _BY_NAME[str(name)]
    except KeyError:
        if _TEMPORAL_64.match(str(name)):
            return _BY_NAME.setdefault(str(name), DataType(str(name), 64, np.int64))
        if _TEMPORAL_32.match(str(name)):
            return _BY_NAME.setdefault(str(name), DataType(str(name), 32, np.int32))
        raise _lib.ArrowNotImplementedError(f"type {name} is not supported by arrow_amd") from None


def type_from_numpy(dt) -> DataType:
    try:
        return _BY_NP[np.dtype(dt)]
    except KeyError:
        raise _lib.ArrowNotImplementedError(f"numpy dtype {dt} is not supported") from None


# ------------------------------------------------------------------ buffers
def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def alloc(nbytes: int, device: torch.device | None = None, zero: bool = False) -> torch.Tensor:
    """A device buffer of at least `nbytes`, padded to 64 bytes (>= 64 so it is never empty)."""
    device = device or default_device()
    n = max(64, _round_up(int(nbytes), 64))
    if zero:
        return torch.zeros(n, dtype=torch.uint8, device=device)
    return torch.empty(n, dtype=torch.uint8, device=device)


def bitmap_nbytes(nbits: int) -> int:
    """Bitmaps are written in whole 64-bit words (include/arrow_amd.h conventions)."""
    return ((int(nbits) + 63) // 64) * 8


def to_device(a: np.ndarray, device: torch.device | None = None) -> torch.Tensor:
    """Copy host bytes into a fresh padded device buffer."""
    device = device or default_device()
    raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    buf = alloc(raw.nbytes, device, zero=True)
    if raw.nbytes:
        buf[: raw.nbytes].copy_(torch.from_numpy(raw.copy()))
    return buf


def pack_validity(valid) -> np.ndarray:
    """bool array (True = valid) -> LSB-first bitmap bytes, padded to 64-bit words."""
    v = np.asarray(valid, dtype=bool)
    out = np.zeros(bitmap_nbytes(len(v)), dtype=np.uint8)
    packed = np.packbits(v, bitorder="little")
    out[: len(packed)] = packed
    return out


def unpack_validity(bitmap: np.ndarray, offset: int, length: int) -> np.ndarray:
    bits = np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")
    return bits[offset: offset + length].astype(bool)


class ListArray:
    """arrow::ListArray twin on the device: buffers = [validity or None, int32 offsets (length + 1)], one child (an Array
    or another ListArray).  `type` is the pyarrow list type (it names the item field).  What the Parquet reader hands
    back for repeated columns."""

    def __init__(self, pa_type, length: int, buffers, child, null_count: int = 0, offset: int = 0):
        self.type = pa_type
        self.length = int(length)
        self.buffers = list(buffers)
        self.child = child
        self.null_count = int(null_count)
        self.offset = int(offset)

    @property
    def device(self) -> torch.device:
        return self.buffers[1].device

    def __len__(self):
        return self.length

    def to_pyarrow(self):
        import pyarrow as pa

        offs = self.buffers[1].cpu().numpy().view(np.int32)[: self.offset + self.length + 1]
        vb = None
        if self.buffers[0] is not None:
            vb = pa.py_buffer(self.buffers[0].cpu().numpy()[: (self.offset + self.length + 7) // 8].tobytes())
        child = self.child.to_pyarrow()
        want = self.type.value_field.type
        if child.type != want:
            child = child.cast(want)
        return pa.Array.from_buffers(self.type, self.length, [vb, pa.py_buffer(offs.tobytes())],
                                     null_count=self.null_count if vb is not None else 0, offset=self.offset,
                                     children=[child])

    def to_pylist(self):
        return self.to_pyarrow().to_pylist()


class StructArray:
    """arrow::StructArray twin on the device: buffers = [validity or None], one child per member.  `type` is the pyarrow
    struct type.  What the Parquet reader hands back for struct columns."""

    def __init__(self, pa_type, length: int, buffers, children, null_count: int = 0):
        self.type = pa_type
        self.length = int(length)
        self.buffers = list(buffers)
        self.children = list(children)
        self.null_count = int(null_count)

    def __len__(self):
        return self.length

    def to_pyarrow(self):
        import pyarrow as pa

        vb = None
        if self.buffers[0] is not None:
            vb = pa.py_buffer(self.buffers[0].cpu().numpy()[: (self.length + 7) // 8].tobytes())
        kids = []
        for j, child in enumerate(self.children):
            arr = child.to_pyarrow()
            want = self.type.field(j).type
            kids.append(arr if arr.type == want else arr.cast(want))
        return pa.Array.from_buffers(self.type, self.length, [vb], null_count=self.null_count if vb is not None else 0,
                                     children=kids)

    def to_pylist(self):
        return self.to_pyarrow().to_pylist()


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------ Array
class Array:
    """arrow::ArrayData twin: {type, length, null_count, offset, buffers=[validity, data]}."""

    def __init__(self, type: DataType, length: int, buffers, null_count: int = kUnknownNullCount,
                 offset: int = 0):
        self.type = type
        self.length = int(length)
        self.buffers = list(buffers)
        self.offset = int(offset)
        if self.buffers[0] is None and null_count == kUnknownNullCount:
            null_count = 0  # ArrayData::GetNullCount with no bitmap
        self._null_count = int(null_count)

    @property
    def null_count(self) -> int:
        """Exact, 0 or kUnknownNullCount; a deferred device-side count is resolved on first read."""
        if callable(self._null_count):
            self._null_count = int(self._null_count())
        return self._null_count

    def set_lazy_null_count(self, fn) -> None:
        self._null_count = fn

    # -- plumbing
    @property
    def device(self) -> torch.device:
        return self.buffers[1].device

    @property
    def validity(self):
        return self.buffers[0]

    @property
    def data(self):
        return self.buffers[1]

    def may_have_nulls(self) -> bool:
        """ArraySpan::MayHaveNulls (array/data.h): null_count != 0 and a bitmap exists."""
        return self.null_count != 0 and self.buffers[0] is not None

    def span(self) -> _lib.ArxSpan:
        """The struct handed across the C ABI (device pointers, no copies)."""
        return _lib.ArxSpan(_ptr(self.buffers[0]), _ptr(self.buffers[1]), self.offset, self.length,
                            self.null_count)

    def binary_span(self) -> _lib.ArxBinarySpan:
        """ArxBinarySpan of a binary / utf8 array (buffers = [validity, offsets, data])."""
        assert is_base_binary(self.type)
        return _lib.ArxBinarySpan(_ptr(self.buffers[0]), _ptr(self.buffers[1]), _ptr(self.buffers[2]),
                                  self.offset, self.length, self.null_count)

    def values_ptr(self) -> int:
        """Device address of logical element 0 (fixed-width, non-boolean types)."""
        assert self.type.bit_width >= 8
        return self.buffers[1].data_ptr() + self.offset * self.type.byte_width

    def slice(self, offset: int, length: int | None = None) -> "Array":
        """Array::Slice: zero-copy, same buffers, shifted offset."""
        offset = int(offset)
        if offset < 0 or offset > self.length:
            raise _lib.ArrowInvalid("slice offset out of range")
        if length is None:
            length = self.length - offset
        length = min(int(length), self.length - offset)
        nc = kUnknownNullCount if self.buffers[0] is not None else 0
        if self.null_count == 0:
            nc = 0
        return Array(self.type, length, self.buffers, nc, self.offset + offset)

    def __len__(self):
        return self.length

    def __repr__(self):
        return (f"<arrow_amd.Array {self.type.name}[{self.length}] offset={self.offset} "
                f"null_count={self.null_count} on {self.device}>")

    # -- construction
    @staticmethod
    def from_numpy(values, valid=None, device=None, type: DataType | None = None) -> "Array":
        """values: numpy array (bool -> boolean bitmap); valid: bool array (True = valid)."""
        values = np.asarray(values)
        t = type or type_from_numpy(values.dtype)
        n = len(values)
        if t == bool_:
            data = to_device(pack_validity(values.astype(bool)), device)
        else:
            data = to_device(values.astype(t.np_dtype, copy=False), device)
        vbuf, nc = None, 0
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            assert len(valid) == n
            vbuf = to_device(pack_validity(valid), device)
            nc = int(n - valid.sum())
        return Array(t, n, [vbuf, data], nc, 0)

    @staticmethod
    def from_pyarrow(arr, device=None) -> "Array":
        """Upload a pyarrow.Array (fixed-width or boolean), preserving offset and bitmaps."""
        t = type_from_name(str(arr.type))
        vb, db = arr.buffers()[0], arr.buffers()[1]
        n, off = len(arr), arr.offset
        if is_base_binary(t):
            offs = np.frombuffer(db, dtype=np.int32)[: off + n + 1] if db is not None else np.zeros(1, np.int32)
            nbytes = int(offs[-1]) if len(offs) else 0
            bb = arr.buffers()[2]
            host = np.frombuffer(bb, dtype=np.uint8)[:nbytes] if bb is not None and nbytes else np.zeros(0, np.uint8)
            vbuf = None
            if vb is not None:
                vbuf = to_device(np.frombuffer(vb, dtype=np.uint8)[: (off + n + 7) // 8], device)
            return Array(t, n, [vbuf, to_device(offs, device), to_device(host, device)],
                         arr.null_count if vb is not None else 0, off)
        if t == bool_:
            need = (off + n + 7) // 8
        else:
            need = (off + n) * t.byte_width
        host = np.frombuffer(db, dtype=np.uint8)[:need] if db is not None and need else np.zeros(0, np.uint8)
        data = to_device(host, device)
        vbuf = None
        if vb is not None:
            vneed = (off + n + 7) // 8
            vbuf = to_device(np.frombuffer(vb, dtype=np.uint8)[:vneed], device)
        return Array(t, n, [vbuf, data], arr.null_count if vb is not None else 0, off)

    # -- export
    def to_numpy(self):
        """Returns (values, valid): numpy values of the logical range and a bool array or None."""
        n, off = self.length, self.offset
        raw = self.buffers[1].cpu().numpy()
        if self.type == bool_:
            values = unpack_validity(raw, off, n)
        else:
            w = self.type.byte_width
            values = raw[off * w: (off + n) * w].view(self.type.np_dtype).copy()
        valid = None
        if self.buffers[0] is not None:
            valid = unpack_validity(self.buffers[0].cpu().numpy(), off, n)
        return values, valid

    def to_pyarrow(self):
        import pyarrow as pa

        if is_base_binary(self.type):
            offs = self.buffers[1].cpu().numpy().view(np.int32)[: self.offset + self.length + 1]
            nbytes = int(offs[-1]) if len(offs) else 0
            data = self.buffers[2].cpu().numpy()[:nbytes] if self.buffers[2] is not None else np.zeros(0, np.uint8)
            vb = None
            if self.buffers[0] is not None:
                vb = pa.py_buffer(self.buffers[0].cpu().numpy()[: (self.offset + self.length + 7) // 8].tobytes())
            pt = pa.binary() if self.type.name == "binary" else pa.string()
            return pa.Array.from_buffers(pt, self.length,
                                         [vb, pa.py_buffer(offs.tobytes()), pa.py_buffer(data.tobytes())],
                                         offset=self.offset)
        values, valid = self.to_numpy()
        mask = None if valid is None else ~valid
        if is_temporal(self.type):
            return pa.array(values, mask=mask).view(_pa_temporal(self.type.name))
        return pa.array(values, type=pa.type_for_alias(_PA_ALIAS[self.type.name]), mask=mask)

    def to_pylist(self):
        if is_base_binary(self.type):
            return self.to_pyarrow().to_pylist()
        values, valid = self.to_numpy()
        if valid is None:
            return values.tolist()
        return [v if ok else None for v, ok in zip(values.tolist(), valid.tolist())]


def _pa_temporal(name: str):
    import pyarrow as pa

    kind, _, rest = name.partition("[")
    unit, _, tz = rest.rstrip("]").partition(", tz=")
    if kind == "timestamp":
        return pa.timestamp(unit, tz=tz or None)
    return {"duration": pa.duration, "time64": pa.time64, "time32": pa.time32}[kind](unit) if kind in ("duration", "time64", "time32") \
        else (pa.date32() if kind == "date32" else pa.date64())


_PA_ALIAS = {"bool": "bool", "int8": "int8", "uint8": "uint8", "int16": "int16", "uint16": "uint16",
             "int32": "int32", "uint32": "uint32", "int64": "int64", "uint64": "uint64",
             "float": "float32", "double": "float64"}


class Scalar:
    """arrow::Scalar twin for broadcast arguments (is_valid + value)."""

    def __init__(self, value, type: DataType, is_valid: bool = True):
        self.value = value
        self.type = type
        self.is_valid = bool(is_valid) and value is not None


def array(obj, type: DataType | None = None, device=None) -> Array:
    """Like pyarrow.array(): python list (None = null) / numpy array -> device Array."""
    if isinstance(obj, Array):
        return obj
    if isinstance(obj, np.ndarray):
        return Array.from_numpy(obj, None, device, type)
    obj = list(obj)
    valid = np.array([x is not None for x in obj], dtype=bool)
    t = type
    if t is None:
        sample = next((x for x in obj if x is not None), 0)
        t = bool_ if isinstance(sample, bool) else (float64 if isinstance(sample, float) else int64)
    fill = False if t == bool_ else 0
    vals = np.array([fill if x is None else x for x in obj], dtype=t.np_dtype)
    if len(obj) == 0:
        vals = np.zeros(0, dtype=t.np_dtype)
    return Array.from_numpy(vals, None if valid.all() else valid, device, t)


class RunEndEncoded:
    """run_end_encoded<run_end_type, value_type> (docs/source/format/Columnar.rst "Run-End Encoded Layout"): two
    children — run_ends (int16 / int32 / int64, strictly increasing, the last one >= offset + length) and values (one
    per run) — plus a logical offset and length.  Only what the hot path needs: boolean values as a filter mask."""

    def __init__(self, run_ends: Array, values: Array, length: int, offset: int = 0):
        if run_ends.type.name not in ("int16", "int32", "int64"):
            raise _lib.ArrowInvalid("run ends must be int16, int32 or int64")
        self.run_ends, self.values, self.length, self.offset = run_ends, values, int(length), int(offset)
        self.type = DataType(f"run_end_encoded<run_ends: {run_ends.type.name}, values: {values.type.name}>", 0, None)

    @property
    def device(self):
        return self.values.device

    @staticmethod
    def from_pyarrow(arr, device=None) -> "RunEndEncoded":
        return RunEndEncoded(Array.from_pyarrow(arr.run_ends, device), Array.from_pyarrow(arr.values, device), len(arr),
                             arr.offset)
