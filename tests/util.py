"""Shared helpers of the parity tests: host-side array descriptions, conversions to the device
arrays / pyarrow / oracle inputs, seeded generators (in the spirit of arrow/testing/random.h)."""
from __future__ import annotations

import dataclasses

import numpy as np

from oracle import oracle as O

try:  # the reference's own build, when the wheel is in the image
    import pyarrow as pa
    import pyarrow.compute as pc
except Exception:  # pragma: no cover
    pa = pc = None

kRandomSeed = 0x0FF1CE  # compute/kernels/test_util_internal.h:119


@dataclasses.dataclass
class HostArray:
    """A (possibly sliced) fixed-width or boolean Arrow array held in numpy pieces."""
    values: np.ndarray            # full buffer incl. `offset` leading elements (bool: bool array)
    valid: np.ndarray | None      # bool array over the full buffer (True = valid) or None
    offset: int
    length: int

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def is_bool(self):
        return self.values.dtype == np.bool_

    # logical views
    def logical_values(self):
        return self.values[self.offset: self.offset + self.length]

    def logical_valid(self):
        if self.valid is None:
            return np.ones(self.length, dtype=bool)
        return self.valid[self.offset: self.offset + self.length]

    # oracle inputs
    def data_bytes(self):
        return O.pack_bits(self.values) if self.is_bool else np.ascontiguousarray(self.values)

    def valid_bitmap(self):
        return None if self.valid is None else O.pack_bits(self.valid)

    def null_count(self):
        return 0 if self.valid is None else int((~self.logical_valid()).sum())

    def to_device(self, amd):
        full = amd.Array.from_numpy(self.values, self.valid)
        if self.offset == 0 and self.length == len(self.values):
            return full
        out = full.slice(self.offset, self.length)
        return out

    def to_pyarrow(self):
        mask = None if self.valid is None else ~self.valid
        arr = pa.array(self.values, mask=mask)
        return arr.slice(self.offset, self.length)


def random_array(rng, dtype, length, null_p=0.0, offset=0, tail=0, lo=None, hi=None) -> HostArray:
    """length logical elements preceded by `offset` and followed by `tail` garbage elements."""
    n = offset + length + tail
    dt = np.dtype(dtype)
    if dt == np.bool_:
        vals = rng.random(n) < 0.5
    elif dt.kind == "f":
        vals = rng.standard_normal(n).astype(dt)
    else:
        info = np.iinfo(dt)
        lo_ = info.min if lo is None else lo
        hi_ = info.max if hi is None else hi
        vals = rng.integers(lo_, hi_, size=n, dtype=dt, endpoint=True)
    valid = None
    if null_p > 0:
        valid = rng.random(n) >= null_p
    return HostArray(vals, valid, offset, length)


def random_mask(rng, length, true_p, null_p=0.0, offset=0, tail=0) -> HostArray:
    n = offset + length + tail
    vals = rng.random(n) < true_p
    valid = (rng.random(n) >= null_p) if null_p > 0 else None
    return HostArray(vals, valid, offset, length)


def device_bitmap_to_bool(buf, length):
    """torch uint8 buffer holding an LSB-first bitmap -> (bool[length], padding_is_zero)."""
    raw = buf.cpu().numpy()
    nbytes = ((length + 63) // 64) * 8
    bits = np.unpackbits(raw[:nbytes], bitorder="little")
    return bits[:length].astype(bool), not bits[length:].any()


def oracle_bitmap_to_bool(bm, length):
    return O.unpack_bits(bm, 0, length)


def first_mismatch(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    bad = np.nonzero(a != b)[0]
    if len(bad) == 0:
        return None
    i = int(bad[0])
    return f"{len(bad)} mismatches, first at {i}: got {a[i]!r} want {b[i]!r} (ctx got {a[max(0,i-2):i+3]} want {b[max(0,i-2):i+3]})"


def assert_equal(got, want, what=""):
    msg = first_mismatch(got, want)
    assert msg is None, f"{what}: {msg}"


@dataclasses.dataclass
class HostBinaryArray:
    """A (possibly sliced) binary / utf8 array: int32 offsets over the full buffer + bytes."""
    offsets: np.ndarray           # int32[offset + length + tail + 1]
    data: np.ndarray              # uint8 bytes
    valid: np.ndarray | None      # bool over the full buffer or None
    offset: int
    length: int
    utf8: bool = False

    def valid_bitmap(self):
        return None if self.valid is None else O.pack_bits(self.valid)

    def logical_valid(self):
        if self.valid is None:
            return np.ones(self.length, dtype=bool)
        return self.valid[self.offset: self.offset + self.length]

    def logical_values(self):
        """The rows as Python bytes (null rows too: whatever their offsets span)."""
        o = self.offsets
        return [self.data[o[i]:o[i + 1]].tobytes() for i in range(self.offset, self.offset + self.length)]

    def null_count(self):
        return 0 if self.valid is None else int((~self.valid[self.offset: self.offset + self.length]).sum())

    def to_pyarrow(self):
        n = len(self.offsets) - 1
        vb = None if self.valid is None else pa.py_buffer(O.pack_bits(self.valid).tobytes())
        arr = pa.Array.from_buffers(pa.string() if self.utf8 else pa.binary(), n,
                                    [vb, pa.py_buffer(self.offsets.tobytes()), pa.py_buffer(self.data.tobytes())])
        return arr.slice(self.offset, self.length)

    def to_device(self, amd):
        A = amd.array
        n = len(self.offsets) - 1
        vbuf = None if self.valid is None else A.to_device(A.pack_validity(self.valid))
        t = A.utf8 if self.utf8 else A.binary
        full = A.Array(t, n, [vbuf, A.to_device(self.offsets), A.to_device(self.data)],
                       0 if self.valid is None else int(n - self.valid.sum()), 0)
        if self.offset == 0 and self.length == n:
            return full
        return full.slice(self.offset, self.length)


def random_binary_pool(rng, length, cardinality, null_p=0.0, offset=0, max_len=24) -> "HostBinaryArray":
    """Strings drawn from a pool of `cardinality` distinct-ish values (group-by keys): shared prefixes, NUL bytes, "" included."""
    pool = [b""] + [bytes(rng.integers(0, 3, size=int(rng.integers(0, max_len, endpoint=True))).astype(np.uint8)) for _ in range(max(1, cardinality))]
    n = offset + length + 2
    pick = rng.integers(0, len(pool), size=n)
    lens = np.array([len(pool[i]) for i in pick], dtype=np.int64)
    offsets = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(pool[i] for i in pick), dtype=np.uint8).copy()
    valid = (rng.random(n) >= null_p) if null_p > 0 else None
    return HostBinaryArray(offsets, data, valid, offset, length)


def random_binary(rng, length, null_p=0.0, offset=0, tail=0, max_len=24, utf8=False, empty_p=0.1) -> HostBinaryArray:
    """RandomArrayGenerator::String-like (arrow/testing/random.h): lengths 0..max_len, some empty."""
    n = offset + length + tail
    lens = rng.integers(0, max_len, size=n, endpoint=True).astype(np.int64)
    lens[rng.random(n) < empty_p] = 0
    offsets = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    if utf8:
        data = rng.integers(0x20, 0x7E, size=total, endpoint=True).astype(np.uint8)
    else:
        data = rng.integers(0, 255, size=total, endpoint=True).astype(np.uint8)
    valid = (rng.random(n) >= null_p) if null_p > 0 else None
    return HostBinaryArray(offsets, data, valid, offset, length, utf8)
