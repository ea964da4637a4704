"""ctypes/numpy front-end of oracle/arx_oracle.c — TEST INFRASTRUCTURE ONLY.

Arrays are described by plain numpy pieces: `values` (1-D numpy array of the physical
type), `valid` (uint8 numpy bitmap bytes or None) and an element `offset`.  Bitmaps are
LSB-first like Arrow's.  See the C file for the reference citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libarx_oracle.so")

INDEX_TYPES = {
    np.dtype("uint8"): 0, np.dtype("int8"): 1, np.dtype("uint16"): 2, np.dtype("int16"): 3,
    np.dtype("uint32"): 4, np.dtype("int32"): 5, np.dtype("uint64"): 6, np.dtype("int64"): 7,
}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "arx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        p, i64, i32, u64, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32
        L.arxo_filter_output_size.restype = i64
        L.arxo_filter_output_size.argtypes = [p, p, i64, i64, i32]
        L.arxo_filter.restype = i64
        L.arxo_filter.argtypes = [p, i32, p, i64, p, p, i64, i64, i32, p, p]
        L.arxo_mask_to_indices.restype = i64
        L.arxo_mask_to_indices.argtypes = [p, p, i64, i64, i32, i32, p, p]
        L.arxo_check_index_bounds.restype = i32
        L.arxo_check_index_bounds.argtypes = [p, i32, p, i64, i64, u64, p, p]
        L.arxo_take.restype = i64
        L.arxo_take.argtypes = [p, i32, p, i64, p, i32, p, i64, i64, p, p]
        L.arxo_binary_take.restype = i64
        L.arxo_binary_take.argtypes = [p, p, p, i64, p, i32, p, i64, i64, p, p, p, p]
        L.arxo_cast_f64_f32.restype = None
        L.arxo_cast_f64_f32.argtypes = [p, i64, p]
        L.arxo_greater_f64.restype = None
        L.arxo_greater_f64.argtypes = [p, i32, p, i32, i64, p]
        L.arxo_greater_i64.restype = None
        L.arxo_greater_i64.argtypes = [p, p, i64, p]
        L.arxo_add_i64.restype = None
        L.arxo_add_i64.argtypes = [p, p, i64, p]
        L.arxo_add_f64.restype = None
        L.arxo_add_f64.argtypes = [p, p, i64, p]
        L.arxo_bitmap_and.restype = None
        L.arxo_bitmap_and.argtypes = [p, i64, p, i64, i64, p]
        L.arxo_bitmap_popcount.restype = i64
        L.arxo_bitmap_popcount.argtypes = [p, i64, i64]
        L.arxo_sort_indices_64.restype = i32
        L.arxo_sort_indices_64.argtypes = [p, p, i64, i64, i32, i32, i32, p]
        L.arxo_sort_indices.restype = i32
        L.arxo_sort_indices.argtypes = [p, i32, p, i64, i64, i32, i32, p]
        L.arxo_groupby_sum_i64.restype = i64
        L.arxo_groupby_sum_i64.argtypes = [p, p, i64, p, p, i64, i64, i32, u32, p, p, p, p, p, p]
        L.arxo_hash_sum_i64_consume.restype = None
        L.arxo_hash_sum_i64_consume.argtypes = [p, p, i64, i32, i64, i32, p, i64, p, p, p]
        L.arxo_hash_sum_i64_merge.restype = None
        L.arxo_hash_sum_i64_merge.argtypes = [p, p, p, p, p, p, p, i64]
        L.arxo_hash_sum_i64_finalize.restype = i64
        L.arxo_hash_sum_i64_finalize.argtypes = [p, p, i64, i32, u32, p]
    return _lib


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def bitmap_bytes(nbits: int) -> int:
    """Bytes of an output bitmap, padded to 64-bit words like the device library."""
    return ((nbits + 63) // 64) * 8


# ---------------------------------------------------------------- bit helpers (numpy)
def pack_bits(bools) -> np.ndarray:
    """bool array -> LSB-first bitmap bytes padded to 8-byte multiples."""
    b = np.asarray(bools, dtype=bool)
    out = np.zeros(bitmap_bytes(len(b)), dtype=np.uint8)
    packed = np.packbits(b, bitorder="little")
    out[: len(packed)] = packed
    return out


def unpack_bits(bitmap: np.ndarray, offset: int, length: int) -> np.ndarray:
    bits = np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")
    return bits[offset: offset + length].astype(bool)


# ---------------------------------------------------------------- kernels
def filter_output_size(mask, mask_valid, mask_off, length, null_selection) -> int:
    return int(lib().arxo_filter_output_size(_ptr(mask), _ptr(mask_valid), mask_off, length,
                                             null_selection))


def filter(values, values_valid, values_off, mask, mask_valid, mask_off, length, null_selection,
           want_validity: bool):
    """Returns (out_values, out_valid_bitmap_or_None)."""
    w = values.dtype.itemsize
    n = filter_output_size(mask, mask_valid, mask_off, length, null_selection)
    out = np.zeros(n, dtype=values.dtype)
    ov = np.zeros(bitmap_bytes(n), dtype=np.uint8) if want_validity else None
    got = lib().arxo_filter(_ptr(values), w, _ptr(values_valid), values_off, _ptr(mask),
                            _ptr(mask_valid), mask_off, length, null_selection, _ptr(out), _ptr(ov))
    assert got == n
    return out, ov


def mask_to_indices(mask, mask_valid, mask_off, length, null_selection, want_validity: bool):
    n = filter_output_size(mask, mask_valid, mask_off, length, null_selection)
    dt = np.uint16 if length <= 65535 else np.uint32
    out = np.zeros(n, dtype=dt)
    ov = np.zeros(bitmap_bytes(n), dtype=np.uint8) if want_validity else None
    got = lib().arxo_mask_to_indices(_ptr(mask), _ptr(mask_valid), mask_off, length, null_selection,
                                     out.dtype.itemsize, _ptr(out), _ptr(ov))
    assert got == n
    return out, ov


def check_index_bounds(indices, idx_valid, idx_off, length, upper_limit):
    """Returns None if in bounds, else the first offending index (python int)."""
    t = INDEX_TYPES[indices.dtype]
    bs, bu = C.c_int64(0), C.c_uint64(0)
    rc = lib().arxo_check_index_bounds(_ptr(indices), t, _ptr(idx_valid), idx_off, length,
                                       upper_limit, C.byref(bs), C.byref(bu))
    if rc == 0:
        return None
    return int(bs.value) if (t & 1) else int(bu.value)


def take(values, values_valid, values_off, indices, idx_valid, idx_off, length,
         want_validity: bool):
    """Returns (out_values, out_valid_bitmap_or_None, valid_count)."""
    w = values.dtype.itemsize
    out = np.zeros(length, dtype=values.dtype)
    ov = np.zeros(bitmap_bytes(length), dtype=np.uint8) if want_validity else None
    vc = lib().arxo_take(_ptr(values), w, _ptr(values_valid), values_off, _ptr(indices),
                         INDEX_TYPES[indices.dtype], _ptr(idx_valid), idx_off, length, _ptr(out),
                         _ptr(ov))
    return out, ov, int(vc)


def binary_take(offsets, data, values_valid, values_off, indices, idx_valid, idx_off, length):
    """Take on binary/utf8 values (int32 offsets).  Returns (out_offsets int32[length+1],
    out_data uint8[total], out_valid bitmap, valid_count)."""
    out_off = np.zeros(length + 1, dtype=np.int32)
    ov = np.zeros(bitmap_bytes(length), dtype=np.uint8)
    vc = C.c_int64(0)
    args = (_ptr(offsets), _ptr(data), _ptr(values_valid), values_off, _ptr(indices),
            INDEX_TYPES[indices.dtype], _ptr(idx_valid), idx_off, length)
    total = lib().arxo_binary_take(*args, _ptr(out_off), None, _ptr(ov), C.byref(vc))
    assert total >= 0, "offset overflow"
    out_data = np.zeros(max(total, 1), dtype=np.uint8)
    lib().arxo_binary_take(*args, _ptr(out_off), _ptr(out_data), _ptr(ov), C.byref(vc))
    return out_off, out_data[:total], ov, int(vc.value)


def binary_filter(offsets, data, values_valid, values_off, mask, mask_valid, mask_off, length, null_selection):
    """Filter on binary/utf8 values == take(GetTakeIndices(mask)) (BinaryFilterImpl)."""
    idx, idx_bm = mask_to_indices(mask, mask_valid, mask_off, length, null_selection, True)
    idx32 = idx.astype(np.uint32)
    return binary_take(offsets, data, values_valid, values_off, idx32, idx_bm, 0, len(idx32))


def cast_i64_i32(values, valid=None, allow_int_overflow=False):
    """CastIntegerToInteger int64 -> int32 (scalar_cast_numeric.cc:46-54): unless allow_int_overflow,
    IntegersInRange (util/int_util.cc:594-665) fails on the first valid slot, in row order, whose value
    is outside [INT32_MIN, INT32_MAX] with "Integer value V not in range: L to U"; then every slot is
    static_cast (two's-complement truncation).  Returns (int32 array, error text or None)."""
    v = np.asarray(values, dtype=np.int64)
    ok = np.ones(len(v), bool) if valid is None else np.asarray(valid, bool)
    err = None
    if not allow_int_overflow:
        for i in range(len(v)):
            if ok[i] and not (-2**31 <= int(v[i]) <= 2**31 - 1):
                err = f"Integer value {int(v[i])} not in range: -2147483648 to 2147483647"
                break
    return v.astype(np.int32), err


def cast_i64_f64(values, valid=None, allow_float_truncate=False):
    """CastIntegerToFloating int64 -> float64 (scalar_cast_numeric.cc:270-279): unless allow_float_truncate,
    the IntegersInRange check with the bounds -2^53 .. 2^53 (:203-227); then static_cast<double> on every slot
    (round-to-nearest-even, as numpy's astype).  Returns (float64 array, error text or None)."""
    v = np.asarray(values, dtype=np.int64)
    ok = np.ones(len(v), bool) if valid is None else np.asarray(valid, bool)
    err = None
    if not allow_float_truncate:
        for i in range(len(v)):
            if ok[i] and not (-2**53 <= int(v[i]) <= 2**53):
                err = f"Integer value {int(v[i])} not in range: -9007199254740992 to 9007199254740992"
                break
    return v.astype(np.float64), err


def cast_f64_f32(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty(len(a), dtype=np.float32)
    lib().arxo_cast_f64_f32(_ptr(a), len(a), _ptr(out))
    return out


def greater_f64(left, right) -> np.ndarray:
    """left/right: float64 arrays or python floats (scalar broadcast).  Returns bitmap bytes."""
    ls, rs = np.isscalar(left), np.isscalar(right)
    la = np.array([left], dtype=np.float64) if ls else np.ascontiguousarray(left, dtype=np.float64)
    ra = np.array([right], dtype=np.float64) if rs else np.ascontiguousarray(right, dtype=np.float64)
    n = len(ra) if ls else len(la)
    out = np.zeros(bitmap_bytes(n), dtype=np.uint8)
    lib().arxo_greater_f64(_ptr(la), int(ls), _ptr(ra), int(rs), n, _ptr(out))
    return out


def greater_i64(left, right) -> np.ndarray:
    la = np.ascontiguousarray(left, dtype=np.int64)
    ra = np.ascontiguousarray(right, dtype=np.int64)
    out = np.zeros(bitmap_bytes(len(la)), dtype=np.uint8)
    lib().arxo_greater_i64(_ptr(la), _ptr(ra), len(la), _ptr(out))
    return out


def add(left, right) -> np.ndarray:
    la, ra = np.ascontiguousarray(left), np.ascontiguousarray(right)
    out = np.empty_like(la)
    if la.dtype == np.int64:
        lib().arxo_add_i64(_ptr(la), _ptr(ra), len(la), _ptr(out))
    else:
        lib().arxo_add_f64(_ptr(la), _ptr(ra), len(la), _ptr(out))
    return out


def arith(op: str, left, right, both_valid=None, dtype=None):
    """add / subtract / multiply on operands of ONE numeric type (arrays or python scalars; `dtype` names it when
    neither side is a typed array — default: int64 / float64 as numpy infers).
    Add / Subtract / Multiply (base_arithmetic_internal.h:45-68,98-120,290-330): integer results wrap around in the
    type's width (computed in the unsigned type there; 16-bit multiplies go through uint32, :303-325, same bits);
    AddChecked / SubtractChecked / MultiplyChecked (:70-96,121-150,341-364) give the same values and additionally
    report an overflow of that type, but only for slots the kernel visits, i.e. where both operands are valid.
    Floats: the IEEE operation in the operand type (float32 stays float32).
    Returns (result, overflowed: bool)."""
    la, ra = np.asarray(left), np.asarray(right)
    if dtype is None:
        dtype = la.dtype if la.ndim else ra.dtype
        if dtype.kind == "f" or la.dtype.kind == "f" or ra.dtype.kind == "f":
            dtype = dtype if dtype.kind == "f" else np.dtype(np.float64)
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        with np.errstate(all="ignore"):
            res = {"add": np.add, "subtract": np.subtract, "multiply": np.multiply}[op](la.astype(dtype), ra.astype(dtype))
        return np.asarray(res, dtype=dtype), False
    n = max(la.size if la.ndim else 1, ra.size if ra.ndim else 1)
    lo = np.broadcast_to(la.astype(object), (n,))
    ro = np.broadcast_to(ra.astype(object), (n,))
    exact = [int(a) + int(b) if op == "add" else int(a) - int(b) if op == "subtract" else int(a) * int(b)
             for a, b in zip(lo, ro)]
    info = np.iinfo(dtype)
    span = 1 << (8 * dtype.itemsize)
    wrapped = np.array([((x - info.min) % span) + info.min for x in exact], dtype=object).astype(dtype)
    ovf = np.array([not (info.min <= x <= info.max) for x in exact], dtype=bool)
    if both_valid is not None:
        ovf &= np.asarray(both_valid, bool)
    return wrapped, bool(ovf.any())


def divide(left, right, both_valid=None, checked=False, dtype=None):
    """divide / divide_checked on operands of ONE numeric type (arrays or python scalars; `dtype` names it when neither
    side is a typed array — default: int64 / float64 as numpy infers): Divide / DivideChecked,
    base_arithmetic_internal.h:366-424, visited only where both operands are valid (ScalarBinaryNotNull).
    Integers: C++ truncating division; zero divisor -> "divide by zero" in both forms; for the SIGNED types min / -1 of
    the type's own width -> 0 unchecked, "overflow" checked (DivideWithOverflowGeneric, util/int_util_overflow.h:124-138).
    Floats: IEEE division in the operand type; the checked form fails on a zero divisor.  The Status is overwritten by
    every failing slot, so the last one names the error.  Returns (result, error message or None); the result at slots
    that are not visited or failed is unspecified (0 here)."""
    la, ra = np.asarray(left), np.asarray(right)
    if dtype is None:
        if la.dtype.kind == "f" or ra.dtype.kind == "f":
            typed = la if (la.ndim and la.dtype.kind == "f") else ra if (ra.ndim and ra.dtype.kind == "f") else None
            dtype = typed.dtype if typed is not None else np.dtype(np.float64)
        else:
            dtype = la.dtype if la.ndim else ra.dtype if ra.ndim else np.dtype(np.int64)
    dtype = np.dtype(dtype)
    is_f = dtype.kind == "f"
    n = max(la.size if la.ndim else 1, ra.size if ra.ndim else 1)
    lo, ro = np.broadcast_to(la, (n,)), np.broadcast_to(ra, (n,))
    visit = np.ones(n, bool) if both_valid is None else np.asarray(both_valid, bool)
    out = np.zeros(n, dtype=dtype)
    signed_min = int(np.iinfo(dtype).min) if dtype.kind == "i" else None
    error = None
    for i in range(n):
        if is_f:
            a, b = dtype.type(lo[i]), dtype.type(ro[i])
            if checked and b == 0.0:
                if visit[i]:
                    error = "divide by zero"
                continue
            with np.errstate(all="ignore"):
                out[i] = a / b
        else:
            a, b = int(lo[i]), int(ro[i])
            if b == 0:
                if visit[i]:
                    error = "divide by zero"
            elif signed_min is not None and a == signed_min and b == -1:
                if checked and visit[i]:
                    error = "overflow"
            else:
                q = abs(a) // abs(b)
                out[i] = q if (a < 0) == (b < 0) else -q
    return out, error


_CMP = {"equal": np.equal, "not_equal": np.not_equal, "greater": np.greater,
        "greater_equal": np.greater_equal, "less": np.less, "less_equal": np.less_equal}


def compare(op: str, left, right) -> np.ndarray:
    """equal / not_equal / greater / greater_equal / less / less_equal on int64 or float64 operands
    (arrays or python scalars): the Call bodies of Equal ... LessEqual, kernels/scalar_compare.cc:38-64
    — C++ `==`, `!=`, `>`, `>=` (less / less_equal are the flipped forms, :436-445), i.e. IEEE for
    floats: any ordered comparison or equality with a NaN is false, not_equal is true.  numpy's
    element-wise operators are the same C operators.  Returns bool[n] computed on every slot."""
    with np.errstate(invalid="ignore"):
        return np.asarray(_CMP[op](left, right), dtype=bool)


def kleene(op: str, l_data, l_valid, r_data, r_valid):
    """and_kleene / or_kleene on boolean arrays given as numpy bool arrays (valid None = no nulls):
    KleeneAndOp / KleeneOrOp, kernels/scalar_boolean.cc:179-196,240-257 — per slot, from the
    truth table: and: false if either side is false, null if the rest involves a null; or: dual.
    Returns (data bool[n], valid bool[n]); data under a null slot is what the reference's word
    formula leaves there (and: 0; or: l_true | r_true)."""
    l_data, r_data = np.asarray(l_data, bool), np.asarray(r_data, bool)
    lv = np.ones(len(l_data), bool) if l_valid is None else np.asarray(l_valid, bool)
    rv = np.ones(len(r_data), bool) if r_valid is None else np.asarray(r_valid, bool)
    lt, lf, rt, rf = lv & l_data, lv & ~l_data, rv & r_data, rv & ~r_data
    if op == "and":
        return lt & rt, lf | rf | (lt & rt)
    return lt | rt, lt | rt | (lf & rf)


def bitmap_and(a, a_off, b, b_off, n) -> np.ndarray:
    out = np.zeros(bitmap_bytes(n), dtype=np.uint8)
    lib().arxo_bitmap_and(_ptr(a), a_off, _ptr(b), b_off, n, _ptr(out))
    return out


def bitmap_popcount(a, off, n) -> int:
    return int(lib().arxo_bitmap_popcount(_ptr(a), off, n))


def sort_indices_64(values, valid, offset, length, descending=False, nulls_at_start=False):
    assert values.dtype in (np.uint64, np.int64)
    out = np.empty(length, dtype=np.uint64)
    rc = lib().arxo_sort_indices_64(_ptr(values), _ptr(valid), offset, length,
                                    int(values.dtype == np.int64), int(descending),
                                    0 if nulls_at_start else 1, _ptr(out))
    assert rc == 0
    return out


SORT_KEY_TYPES = {np.dtype("uint64"): 0, np.dtype("int64"): 1, np.dtype("uint32"): 2, np.dtype("int32"): 3,
                  np.dtype("float64"): 4, np.dtype("float32"): 5}


def sort_indices(values, valid, offset, length, descending=False, nulls_at_start=False):
    """array_sort_indices for any supported key type (NaNs next to the nulls, -0.0 == 0.0)."""
    out = np.empty(length, dtype=np.uint64)
    rc = lib().arxo_sort_indices(_ptr(values), SORT_KEY_TYPES[values.dtype], _ptr(valid), offset, length,
                                 int(descending), 0 if nulls_at_start else 1, _ptr(out))
    assert rc == 0
    return out


def sort_indices_multi(keys, descending=None, nulls_at_start=False):
    """SortIndices(Table) with several sort keys (TableSorter, kernels/vector_sort.cc:850-954; row comparison
    MultipleKeyComparator, vector_sort_internal.h): rows ordered lexicographically, key by key, each key with
    its own direction and null placement (SortKey::null_placement, compute/ordering.h:50-61; one bool = the
    same for all keys); within one key nulls then NaNs sit at the far end (at_end) or NaNs-after-nulls at the
    front (at_start) regardless of the direction; ties keep their input order.  `keys` = [(values, valid
    bool array or None)] over the logical rows.  numpy restatement: lexsort (stable) over, per key,
    (class: value / NaN / null, dense rank of the value negated for descending)."""
    descending = list(descending) if descending is not None else [False] * len(keys)
    at_start = [bool(nulls_at_start)] * len(keys) if isinstance(nulls_at_start, (bool, np.bool_)) else list(nulls_at_start)
    columns = []
    for (values, valid), desc, nulls_at_start in zip(keys, descending, at_start):
        values = np.asarray(values)
        n = len(values)
        is_null = np.zeros(n, bool) if valid is None else ~np.asarray(valid, bool)
        is_nan = np.isnan(values) & ~is_null if values.dtype.kind == "f" else np.zeros(n, bool)
        plain = np.where(is_null | is_nan, values.dtype.type(0), values)
        _, rank = np.unique(plain, return_inverse=True)       # (-0.0 and 0.0 share a rank: they compare equal)
        rank = rank.astype(np.int64)
        rank = np.where(is_null | is_nan, 0, -rank if desc else rank)
        cls = np.where(is_null, 0 if nulls_at_start else 2, np.where(is_nan, 1, 2 if nulls_at_start else 0))
        columns += [cls, rank]
    return np.lexsort(tuple(reversed(columns))).astype(np.uint64)


# Wichura (1988), Algorithm AS 241 (PPND16), Applied Statistics 37(3): the published coefficients, highest power first
_PPF_A = (2.5090809287301226727e3, 3.3430575583588128105e4, 6.7265770927008700853e4, 4.5921953931549871457e4,
          1.3731693765509461125e4, 1.9715909503065514427e3, 1.3314166789178437745e2, 3.3871328727963666080e0)
_PPF_B = (5.2264952788528545610e3, 2.8729085735721942674e4, 3.9307895800092710610e4, 2.1213794301586595867e4,
          5.3941960214247511077e3, 6.8718700749205790830e2, 4.2313330701600911252e1, 1.0)
_PPF_C = (7.74545014278341407640e-4, 2.27238449892691845833e-2, 2.41780725177450611770e-1, 1.27045825245236838258e0,
          3.64784832476320460504e0, 5.76949722146069140550e0, 4.63033784615654529590e0, 1.42343711074968357734e0)
_PPF_D = (1.05075007164441684324e-9, 5.47593808499534494600e-4, 1.51986665636164571966e-2, 1.48103976427480074590e-1,
          6.89767334985100004550e-1, 1.67638483018380384940e0, 2.05319162663775882187e0, 1.0)
_PPF_E = (2.01033439929228813265e-7, 2.71155556874348757815e-5, 1.24266094738807843860e-3, 2.65321895265761230930e-2,
          2.96560571828504891230e-1, 1.78482653991729133580e0, 5.46378491116411436990e0, 6.65790464350110377720e0)
_PPF_F = (2.04426310338993978564e-15, 1.42151175831644588870e-7, 1.84631831751005468180e-5, 7.86869131145613259100e-4,
          1.48753612908506148525e-2, 1.36929880922735805310e-1, 5.99832206555887937690e-1, 1.0)


def normal_ppf(p):
    """arrow::internal::NormalPPF (cpp/src/arrow/util/math_internal.cc:26-137): Wichura's AS 241 — a rational function of
    0.180625 - q^2 in the centre (|q = p - 1/2| < 0.425), of sqrt(-log(min(p, 1 - p))) in the tails, Horner form in the
    reference's order of operations (no fused multiply-add: numpy rounds every product and every sum); -inf / +inf at 0 / 1."""
    p = np.asarray(p, np.float64)

    def horner(coef, r):
        acc = np.full_like(r, coef[0])
        for c in coef[1:]:
            acc = acc * r + c
        return acc

    q = p - 0.5
    with np.errstate(divide="ignore", invalid="ignore"):
        rc = 0.180625 - q * q
        centre = q * horner(_PPF_A, rc) / horner(_PPF_B, rc)
        r = np.sqrt(-np.log(np.where(q < 0.0, p, 1.0 - p)))
        mid = horner(_PPF_C, r - 1.6) / horner(_PPF_D, r - 1.6)
        far = horner(_PPF_E, r - 5.0) / horner(_PPF_F, r - 5.0)
        tail = np.copysign(np.where(r < 5.0, mid, far), q)
    out = np.where(np.abs(q) < 0.425, centre, tail)
    out = np.where(p == 0.0, -np.inf, np.where(p == 1.0, np.inf, out))
    return out


def rank(values, valid, descending=False, nulls_at_start=False, tiebreaker="first"):
    """compute "rank" / "rank_quantile" (kernels/vector_rank.cc): the array sorter's order (sort_indices_multi with one
    key: values, NaNs, nulls at_end — or reversed at_start —, ties in row order), MarkDuplicates (:40-72: an index is a
    duplicate when its value equals the one before it; every NaN after the first and every null after the first are
    duplicates), then OrdinalRanker::CreateRankings (:203-263) for min / max / first / dense and
    BaseQuantileRanker::CreateRankings (:163-196) for "quantile".  `values` over the logical rows, `valid` a bool array
    or None.  uint64 ranks (float64 for "quantile" and for "normal" = rank_normal)."""
    values = np.asarray(values)
    n = len(values)
    order = sort_indices_multi([(values, valid)], [descending], nulls_at_start).astype(np.int64)
    is_null = np.zeros(n, bool) if valid is None else ~np.asarray(valid, bool)
    is_nan = np.isnan(values) & ~is_null if values.dtype.kind == "f" else np.zeros(n, bool)
    cls = np.where(is_null, 2, np.where(is_nan, 1, 0))[order]
    plain = np.where(is_null | is_nan, values.dtype.type(0), values)[order]
    dup = np.zeros(n, bool)
    if n > 1:
        dup[1:] = (cls[1:] == cls[:-1]) & ((cls[1:] != 0) | (plain[1:] == plain[:-1]))
    pos = np.arange(n, dtype=np.int64)
    starts = np.where(~dup, pos, 0)
    run_start = np.maximum.accumulate(starts) if n else starts              # position of the run's first row
    nxt = np.where(~dup, pos, n)
    run_end = np.empty(n, np.int64)                                          # position after the run's last row
    if n:
        run_end[:-1] = np.minimum.accumulate(nxt[::-1])[::-1][1:]
        run_end[-1] = n
    if tiebreaker in ("quantile", "normal"):      # "normal": NormalRanker::TransformValue (:204-209) of the quantile rank
        out = np.empty(n, np.float64)
        quantile = (run_start + 0.5 * (run_end - run_start)) / float(n) if n else np.zeros(0)
        out[order] = normal_ppf(quantile) if tiebreaker == "normal" else quantile
        return out
    ranks = {"first": pos + 1, "min": run_start + 1, "max": run_end, "dense": np.cumsum(~dup)}[tiebreaker]
    out = np.empty(n, np.uint64)
    out[order] = ranks.astype(np.uint64)
    return out


def groupby_sum_i64(keys, key_valid, key_off, values, val_valid, val_off, length,
                    skip_nulls=True, min_count=1):
    """Returns dict(keys, key_is_valid, sums, counts, no_nulls, valid) in first-occurrence order."""
    n = max(length, 1)
    ok = np.zeros(n, dtype=np.int32)
    okv = np.zeros(n, dtype=np.uint8)
    os_ = np.zeros(n, dtype=np.int64)
    oc = np.zeros(n, dtype=np.int64)
    onn = np.zeros(n, dtype=np.uint8)
    ov = np.zeros(n, dtype=np.uint8)
    ng = lib().arxo_groupby_sum_i64(_ptr(keys), _ptr(key_valid), key_off, _ptr(values),
                                    _ptr(val_valid), val_off, length, int(skip_nulls), min_count,
                                    _ptr(ok), _ptr(okv), _ptr(os_), _ptr(oc), _ptr(onn), _ptr(ov))
    assert ng >= 0
    return dict(keys=ok[:ng], key_is_valid=okv[:ng], sums=os_[:ng], counts=oc[:ng],
                no_nulls=onn[:ng], valid=ov[:ng])


def groupby_mean_i64(keys, key_valid, key_off, values, val_valid, val_off, length, skip_nulls=True, min_count=1):
    """GroupedMeanImpl<Int64Type> (kernels/hash_aggregate_numeric.cc:352-430) over an int32 key, row at a time:
    the accumulator is a DOUBLE (GroupedMeanAccType :352-356), Consume adds static_cast<double>(v) in row order
    (Reduce :371-375 through GroupedReducingAggregator::Consume :70-83), Finish :402-425 writes
    reduced / count where count >= min_count (else 0 and null), and with !skip_nulls nulls groups that saw a null.
    Groups in first-occurrence order; the null key is one group.  Returns dict(keys, key_is_valid, means, valid)."""
    kv = unpack_bits(key_valid, key_off, length) if key_valid is not None else np.ones(length, bool)
    vv = unpack_bits(val_valid, val_off, length) if val_valid is not None else np.ones(length, bool)
    k = np.asarray(keys)[key_off: key_off + length]
    v = np.asarray(values)[val_off: val_off + length]
    index, out_keys, out_kv, acc, cnt, nn = {}, [], [], [], [], []
    for i in range(length):
        gkey = int(k[i]) if kv[i] else None
        g = index.get(gkey)
        if g is None:
            g = index[gkey] = len(out_keys)
            out_keys.append(0 if gkey is None else gkey)
            out_kv.append(0 if gkey is None else 1)
            acc.append(np.float64(0.0)); cnt.append(0); nn.append(True)
        if vv[i]:
            acc[g] = np.float64(acc[g] + np.float64(v[i]))
            cnt[g] += 1
        else:
            nn[g] = False
    means, valid = [], []
    with np.errstate(invalid="ignore", divide="ignore"):
        for g in range(len(out_keys)):
            ok = cnt[g] >= min_count
            means.append(np.float64(acc[g]) / np.float64(cnt[g]) if ok else np.float64(0.0))
            valid.append(ok and (skip_nulls or nn[g]))
    return dict(keys=np.array(out_keys, dtype=np.int32), key_is_valid=np.array(out_kv, dtype=np.uint8),
                means=np.array(means, dtype=np.float64), valid=np.array(valid, dtype=np.uint8))


def groupby_minmax_i64(keys, key_valid, key_off, values, val_valid, val_off, length, skip_nulls=True):
    """GroupedMinMaxImpl<Int64Type> (kernels/hash_aggregate.cc:330-419) over an int32 key:
    Resize :346-354 (mins = max(), maxes = lowest(), has_values = has_nulls = false), Consume
    :356-369 (valid value -> min/max + has_values; null value -> has_nulls), Finalize :401-419
    (valid = has_values, and with !skip_nulls also !has_nulls; min_count is not consulted).
    Row-at-a-time, groups in first-occurrence order; the null key is one group.
    Returns dict(keys, key_is_valid, mins, maxs, valid)."""
    kv = unpack_bits(key_valid, key_off, length) if key_valid is not None else np.ones(length, bool)
    vv = unpack_bits(val_valid, val_off, length) if val_valid is not None else np.ones(length, bool)
    k = np.asarray(keys)[key_off: key_off + length]
    v = np.asarray(values)[val_off: val_off + length]
    index, out_keys, out_kv = {}, [], []
    mins, maxs, has_values, has_nulls = [], [], [], []
    i64 = np.iinfo(np.int64)
    for i in range(length):
        gkey = int(k[i]) if kv[i] else None
        g = index.get(gkey)
        if g is None:
            g = index[gkey] = len(out_keys)
            out_keys.append(0 if gkey is None else gkey)
            out_kv.append(0 if gkey is None else 1)
            mins.append(i64.max); maxs.append(i64.min); has_values.append(False); has_nulls.append(False)
        if vv[i]:
            x = int(v[i])
            mins[g] = min(mins[g], x); maxs[g] = max(maxs[g], x); has_values[g] = True
        else:
            has_nulls[g] = True
    valid = [hv and (skip_nulls or not hn) for hv, hn in zip(has_values, has_nulls)]
    return dict(keys=np.array(out_keys, np.int32), key_is_valid=np.array(out_kv, np.uint8),
                mins=np.array(mins, np.int64), maxs=np.array(maxs, np.int64), valid=np.array(valid, np.uint8))


class Grouper:
    """arrow::compute::Grouper as GrouperImpl implements it (compute/row/grouper.cc:335-553): every key row is
    encoded to bytes (one encoder per column, a null is part of the encoding: KeyEncoder::kNullByte / kValidByte
    prefix, row/grouper.cc:60-120), an unordered_map from encoded row to group id hands out ids in order of first
    appearance (VisitKeys :405-447: `num_groups_++` on insertion), Lookup (:126 of grouper.h) maps unseen rows to a
    null id whose slot holds 0 (:507-510), GetUniques (:527-553) decodes the stored rows in id order.
    Columns are (values ndarray, valid bool ndarray | None) pairs; floats are compared by their bytes like the encoder
    does (so -0.0 != 0.0 and NaNs with equal payloads are one key, grouper_test.cc:898-910)."""

    def __init__(self, num_keys: int):
        self.num_keys = num_keys
        self.index = {}
        self.rows = []

    @staticmethod
    def _encode(columns, i):
        out = []
        for values, valid in columns:
            if valid is not None and not valid[i]:
                out.append(None)
            else:
                out.append(values[i:i + 1].tobytes())
        return tuple(out)

    def consume(self, columns, insert=True):
        n = len(columns[0][0])
        ids = np.zeros(n, dtype=np.uint32)
        found = np.ones(n, dtype=bool)
        for i in range(n):
            row = self._encode(columns, i)
            g = self.index.get(row)
            if g is None:
                if insert:
                    g = self.index[row] = len(self.rows)
                    self.rows.append(row)
                else:
                    g, found[i] = 0, False
            ids[i] = g
        return ids if insert else (ids, found)

    def lookup(self, columns):
        return self.consume(columns, insert=False)

    @property
    def num_groups(self):
        return len(self.rows)

    def uniques(self, dtypes):
        """One (values, valid) pair per key column, in id order."""
        out = []
        for j, dt in enumerate(dtypes):
            vals = np.zeros(len(self.rows), dtype=dt)
            valid = np.ones(len(self.rows), dtype=bool)
            for g, row in enumerate(self.rows):
                if row[j] is None:
                    valid[g] = False
                else:
                    vals[g] = np.frombuffer(row[j], dtype=dt)[0]
            out.append((vals, valid))
        return out


def grouper_ids_one_batch(columns):
    """The ids Grouper.consume gives a fresh Grouper for ONE batch, vectorised (for batches too long for the
    row-at-a-time restatement above, which tests/test_oracle_pin.py holds it equal to): rows as fixed-size byte
    records [null flags | column bytes with nulls zeroed], np.unique for the distinct records, ids renumbered by the
    first row of every record.  Returns (ids uint32, first_rows int64 in id order)."""
    n = len(columns[0][0])
    if n == 0:
        return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.int64)
    parts = []
    for values, valid in columns:
        v = np.ascontiguousarray(values).copy()
        flag = np.zeros(n, dtype=np.uint8)
        if valid is not None:
            v[~valid] = 0
            flag = (~valid).astype(np.uint8)
        parts.append(flag.reshape(n, 1))
        parts.append(v.view(np.uint8).reshape(n, -1))
    rec = np.ascontiguousarray(np.concatenate(parts, axis=1))
    key = rec.view(np.dtype((np.void, rec.shape[1]))).reshape(n)
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(first), dtype=np.int64)
    rank[order] = np.arange(len(first))
    return rank[inverse.reshape(n)].astype(np.uint32), first[order].astype(np.int64)


def unique_i32(values, valid_bitmap, offset, length, with_counts=False):
    """UniqueAction / ValueCountsAction over RegularHashKernel (kernels/vector_hash.cc:65-120,274-330):
    distinct values in order of first appearance; all nulls are one entry, placed where the first null
    appeared; value_counts also counts every occurrence (nulls included).  Row-at-a-time.
    Returns (values int32[g], is_valid bool[g][, counts int64[g]])."""
    valid = unpack_bits(valid_bitmap, offset, length) if valid_bitmap is not None else np.ones(length, bool)
    v = np.asarray(values)[offset: offset + length]
    index, out, out_valid, counts = {}, [], [], []
    for i in range(length):
        key = int(v[i]) if valid[i] else None
        g = index.get(key)
        if g is None:
            g = index[key] = len(out)
            out.append(0 if key is None else key)
            out_valid.append(key is not None)
            counts.append(0)
        counts[g] += 1
    res = (np.array(out, np.int32), np.array(out_valid, bool))
    return res + (np.array(counts, np.int64),) if with_counts else res


def dictionary_encode_i32(values, valid_bitmap, offset, length, encode_nulls=False):
    """DictEncodeAction over RegularHashKernel (kernels/vector_hash.cc:173-270): index of every row in
    the dictionary of distinct values (first-appearance order).  MASK (default): a null row gets a null
    index and the null is not in the dictionary; ENCODE: the null is a dictionary entry.
    Returns (indices int32[n], index_valid bool[n], dict_values int32[g], dict_valid bool[g])."""
    valid = unpack_bits(valid_bitmap, offset, length) if valid_bitmap is not None else np.ones(length, bool)
    v = np.asarray(values)[offset: offset + length]
    index, dvals, dvalid = {}, [], []
    idx = np.zeros(length, np.int32)
    idx_valid = np.ones(length, bool)
    for i in range(length):
        key = int(v[i]) if valid[i] else None
        if key is None and not encode_nulls:
            idx_valid[i] = False
            continue
        g = index.get(key)
        if g is None:
            g = index[key] = len(dvals)
            dvals.append(0 if key is None else key)
            dvalid.append(key is not None)
        idx[i] = g
    return idx, idx_valid, np.array(dvals, np.int32), np.array(dvalid, bool)


def rle_hybrid_encode(values, bit_width: int) -> bytes:
    """A small writer of the RLE / bit-packed hybrid (rle_encoding_internal.h:40-90) for tests: runs of
    >= 8 equal values become repeated runs, everything else literal runs of 8-value groups."""
    v = [int(x) for x in values]
    out = bytearray()

    def varint(x):
        while x >= 0x80:
            out.append((x & 0x7F) | 0x80)
            x >>= 7
        out.append(x)

    i, n, lit = 0, len(v), []

    def flush_literals():
        nonlocal lit
        while lit:
            groups = min((len(lit) + 7) // 8, 63)
            chunk, lit = lit[: groups * 8], lit[groups * 8:]
            chunk = chunk + [0] * (groups * 8 - len(chunk))
            varint((groups << 1) | 1)
            acc, nbits = 0, 0
            for x in chunk:
                acc |= x << nbits
                nbits += bit_width
            out.extend(acc.to_bytes(groups * bit_width, "little"))

    while i < n:
        j = i
        while j < n and v[j] == v[i]:
            j += 1
        if j - i >= 8 and len(lit) % 8 == 0:
            flush_literals()
            varint((j - i) << 1)
            out.extend(v[i].to_bytes((bit_width + 7) // 8, "little"))
            i = j
        else:
            lit.append(v[i])
            i += 1
    flush_literals()
    return bytes(out)


def rle_hybrid_decode(data: bytes, runs, bit_width: int, num_values: int) -> np.ndarray:
    """Value i of an RLE / bit-packed hybrid block given its run table (out_start, kind, payload):
    repeated run -> the payload; literal run -> bits [k*bit_width, (k+1)*bit_width) after the run's
    first byte, LSB first (RleBitPackedDecoder + BitReader, rle_encoding_internal.h, bit_stream_utils_internal.h).
    Row-at-a-time restatement of what arx_rle_decode_u32 computes."""
    out = np.zeros(num_values, dtype=np.uint32)
    big = int.from_bytes(data, "little")
    starts = [int(r["out_start"]) for r in runs]
    import bisect

    for i in range(num_values):
        r = runs[bisect.bisect_right(starts, i) - 1]
        if int(r["kind"]) == 0:
            out[i] = int(r["payload"])
        else:
            bit = int(r["payload"]) * 8 + (i - int(r["out_start"])) * bit_width
            out[i] = (big >> bit) & ((1 << bit_width) - 1)
    return out


def def_rep_levels_to_list(def_levels, rep_levels, def_level: int, rep_level: int, repeated_ancestor_def_level: int = 0):
    """DefRepLevelsToListInfo (cpp/src/parquet/level_conversion.cc:40-124), one level slot at a time as the reference
    walks them: a slot below the repeated ancestor's level or above this list's repetition level is skipped (:52-55), a
    slot at the list's repetition level continues the current entry (:57-66), anything else starts one (:67-107) — with
    an element when def >= def_level, valid when def >= def_level - 1.  Returns (offsets, validity bools, null_count)."""
    offsets, valid = [0], []
    for d, r in zip(def_levels, rep_levels):
        if d < repeated_ancestor_def_level or r > rep_level:
            continue
        if r == rep_level:
            offsets[-1] += 1
        else:
            offsets.append(offsets[-1] + (1 if d >= def_level else 0))
            valid.append(d >= def_level - 1)
    return np.asarray(offsets, np.int32), np.asarray(valid, bool), int(len(valid) - sum(valid))


def def_levels_to_bitmap(def_levels, def_level: int, repeated_ancestor_def_level: int = 0, has_repeated_parent: bool = False):
    """DefLevelsToBitmapSimd (cpp/src/parquet/level_conversion_inc.h:296-353): with a repeated parent only the slots
    whose def level reaches the ancestor's exist; a slot is valid when def >= def_level.  Returns validity bools."""
    d = np.asarray(def_levels)
    if has_repeated_parent:
        d = d[d >= repeated_ancestor_def_level]
    return d >= def_level


class HashSumState:
    """GroupedReducingAggregator<Int64Type, GroupedSumImpl> with dense group ids:
    resize / consume / merge / finalize (hash_aggregate_numeric.cc:61-152)."""

    def __init__(self, skip_nulls=True, min_count=1):
        self.skip_nulls, self.min_count = bool(skip_nulls), int(min_count)
        self.sums = np.zeros(0, np.int64)
        self.counts = np.zeros(0, np.int64)
        self.no_nulls = np.zeros(0, np.uint8)

    @property
    def num_groups(self):
        return len(self.sums)

    def resize(self, n):
        add = n - self.num_groups
        self.sums = np.concatenate([self.sums, np.zeros(add, np.int64)])
        self.counts = np.concatenate([self.counts, np.zeros(add, np.int64)])
        self.no_nulls = np.concatenate([self.no_nulls, np.ones(add, np.uint8)])

    def consume(self, values, val_valid, val_off, group_ids, scalar=None):
        """scalar: None, or (value, is_valid) to broadcast instead of `values`."""
        gids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        if scalar is None:
            lib().arxo_hash_sum_i64_consume(_ptr(values), _ptr(val_valid), val_off, 0, 0, 0,
                                            _ptr(gids), len(gids), _ptr(self.sums),
                                            _ptr(self.counts), _ptr(self.no_nulls))
        else:
            lib().arxo_hash_sum_i64_consume(None, None, 0, 1, int(scalar[0]), int(scalar[1]),
                                            _ptr(gids), len(gids), _ptr(self.sums),
                                            _ptr(self.counts), _ptr(self.no_nulls))

    def merge(self, other: "HashSumState", mapping):
        m = np.ascontiguousarray(mapping, dtype=np.uint32)
        lib().arxo_hash_sum_i64_merge(_ptr(self.sums), _ptr(self.counts), _ptr(self.no_nulls),
                                      _ptr(other.sums), _ptr(other.counts), _ptr(other.no_nulls),
                                      _ptr(m), len(m))

    def finalize(self):
        """(sums int64[G], valid bool[G], null_count)."""
        v = np.zeros(max(self.num_groups, 1), np.uint8)
        nulls = lib().arxo_hash_sum_i64_finalize(_ptr(self.counts), _ptr(self.no_nulls),
                                                 self.num_groups, int(self.skip_nulls),
                                                 self.min_count, _ptr(v))
        return self.sums.copy(), v[: self.num_groups].astype(bool), int(nulls)


def _valid_rows(val_valid, val_off, n):
    """bool[n] from a validity bitmap (None: all valid) at bit offset val_off."""
    if val_valid is None:
        return np.ones(n, bool)
    bits = np.unpackbits(np.frombuffer(np.ascontiguousarray(val_valid), np.uint8), bitorder="little")
    return bits[val_off:val_off + n].astype(bool)


class HashMinMaxState:
    """GroupedMinMaxImpl<Int64Type> with dense group ids (kernels/hash_aggregate.cc:330-419): Resize :343-353 (new
    groups start at the anti-extrema, no value / no null seen), Consume :355-379, Merge :381-399, Finalize :401-419
    (a group is null when it saw no value, or — !skip_nulls — saw a null; min_count is not consulted).  Plain numpy."""

    def __init__(self, skip_nulls=True, dtype=np.int64):
        """dtype float32 / float64: MinMaxOp = fmin / fmax over NaN anti-extrema (hash_aggregate.cc:306-326) — NaN rows
        are skipped, a group of NaNs only ends as NaN; fmin(+0.0, -0.0) depends on the row order in the reference
        (whichever zero came last), so callers compare zeros numerically."""
        self.skip_nulls = bool(skip_nulls)
        self.dtype = np.dtype(dtype)
        self.is_float = self.dtype.kind == "f"
        self.mins = np.zeros(0, self.dtype)
        self.maxs = np.zeros(0, self.dtype)
        self.has_values = np.zeros(0, bool)
        self.has_nulls = np.zeros(0, bool)

    @property
    def num_groups(self):
        return len(self.mins)

    def resize(self, n):
        add = n - self.num_groups
        if self.is_float:
            self.mins = np.concatenate([self.mins, np.full(add, np.nan, self.dtype)])
            self.maxs = np.concatenate([self.maxs, np.full(add, np.nan, self.dtype)])
            self.has_values = np.concatenate([self.has_values, np.zeros(add, bool)])
            self.has_nulls = np.concatenate([self.has_nulls, np.zeros(add, bool)])
            return
        self.mins = np.concatenate([self.mins, np.full(add, np.iinfo(np.int64).max, np.int64)])
        self.maxs = np.concatenate([self.maxs, np.full(add, np.iinfo(np.int64).min, np.int64)])
        self.has_values = np.concatenate([self.has_values, np.zeros(add, bool)])
        self.has_nulls = np.concatenate([self.has_nulls, np.zeros(add, bool)])

    def consume(self, values, val_valid, val_off, group_ids, scalar=None):
        gids = np.asarray(group_ids, dtype=np.int64)
        n = len(gids)
        if scalar is None:
            ok = _valid_rows(val_valid, val_off, n)
            v = np.asarray(values)[val_off:val_off + n]
        else:
            ok = np.full(n, bool(scalar[1]))
            v = np.full(n, scalar[0], self.dtype)
        lo, hi = (np.fmin, np.fmax) if self.is_float else (np.minimum, np.maximum)
        lo.at(self.mins, gids[ok], v[ok])
        hi.at(self.maxs, gids[ok], v[ok])
        self.has_values[gids[ok]] = True
        self.has_nulls[gids[~ok]] = True

    def merge(self, other: "HashMinMaxState", mapping):
        m = np.asarray(mapping, dtype=np.int64)
        lo, hi = (np.fmin, np.fmax) if self.is_float else (np.minimum, np.maximum)
        lo.at(self.mins, m, other.mins)
        hi.at(self.maxs, m, other.maxs)
        np.logical_or.at(self.has_values, m, other.has_values)
        np.logical_or.at(self.has_nulls, m, other.has_nulls)

    def finalize(self):
        """(mins, maxs, valid bool[G])."""
        valid = self.has_values.copy()
        if not self.skip_nulls:
            valid &= ~self.has_nulls
        return self.mins.copy(), self.maxs.copy(), valid


def sum_float_pairwise(values, valid=None):
    """SumArray for floating point (kernels/aggregate_internal.h:155-232) -> (sum: float64, count): the valid values of
    every run of valid slots in blocks of 16 (a run's last block may be shorter), each block added left to right
    from 0.0 in double, the block sums merged by the binary counter of :170-198 and the left-over levels folded from
    the lowest up (:226-231).  `valid`: bool[n] or None.  numpy for the blocks (np.add.at applies its updates in index
    order, i.e. left to right), a Python loop over the block sums for the counter."""
    v = np.asarray(values).astype(np.float64)
    n = len(v)
    ok = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
    count = int(ok.sum())
    if count == 0:
        return 0.0, 0
    idx = np.flatnonzero(ok)
    prev_ok = np.concatenate([[False], ok[:-1]])
    run_start = np.flatnonzero(ok & ~prev_ok)                       # first row of every run
    start_of = run_start[np.searchsorted(run_start, idx, side="right") - 1]
    pos = idx - start_of                                            # position inside the run
    is_block_start = (pos % 16) == 0
    block_id = np.cumsum(is_block_start) - 1
    block_sums = np.zeros(int(block_id[-1]) + 1, np.float64)
    np.add.at(block_sums, block_id, v[idx])
    levels = max(1, int(count).bit_length()) + 1
    sums = [0.0] * (levels + 1)
    mask = 0
    root = 0
    for b in block_sums.tolist():
        cur = 0
        sums[0] += b
        mask ^= 1
        while (mask >> cur) & 1 == 0:
            b = sums[cur]
            sums[cur] = 0.0
            cur += 1
            sums[cur] += b
            mask ^= 1 << cur
        root = max(root, cur)
    for i in range(1, root + 1):
        sums[i] += sums[i - 1]
    return sums[root], count


class HashCountState:
    """GroupedCountImpl with dense group ids (kernels/hash_aggregate.cc:107-212): mode "only_valid" / "only_null" /
    "all" (CountOptions::CountMode); Merge adds (:156-170); the result is never null.  Plain numpy."""

    def __init__(self, mode="only_valid"):
        self.mode = mode
        self.counts = np.zeros(0, np.int64)

    @property
    def num_groups(self):
        return len(self.counts)

    def resize(self, n):
        self.counts = np.concatenate([self.counts, np.zeros(n - self.num_groups, np.int64)])

    def consume(self, val_valid, val_off, group_ids, scalar_valid=None):
        gids = np.asarray(group_ids, dtype=np.int64)
        n = len(gids)
        ok = _valid_rows(val_valid, val_off, n) if scalar_valid is None else np.full(n, bool(scalar_valid))
        take = np.ones(n, bool) if self.mode == "all" else (ok if self.mode == "only_valid" else ~ok)
        np.add.at(self.counts, gids[take], 1)

    def merge(self, other: "HashCountState", mapping):
        np.add.at(self.counts, np.asarray(mapping, dtype=np.int64), other.counts)


class HashBoolState:
    """GroupedBooleanAggregator<GroupedAnyImpl / GroupedAllImpl> with dense group ids (kernels/hash_aggregate.cc:1232-1398),
    restated as the reference keeps it: reduced (starts at NullValue: False for any, True for all), no_nulls, counts;
    Consume :1250-1296, Merge :1298-1319, Finalize :1321-1350 with AdjustForMinCount :1376-1398.  Plain numpy."""

    def __init__(self, is_all: bool, skip_nulls: bool = True, min_count: int = 1):
        self.is_all, self.skip_nulls, self.min_count = bool(is_all), bool(skip_nulls), int(min_count)
        self.reduced = np.zeros(0, bool)
        self.no_nulls = np.zeros(0, bool)
        self.counts = np.zeros(0, np.int64)

    @property
    def num_groups(self):
        return len(self.counts)

    def resize(self, n):
        add = n - self.num_groups
        self.reduced = np.concatenate([self.reduced, np.full(add, self.is_all)])
        self.no_nulls = np.concatenate([self.no_nulls, np.ones(add, bool)])
        self.counts = np.concatenate([self.counts, np.zeros(add, np.int64)])

    def _update(self, gids, values):
        if self.is_all:
            self.reduced[gids[~values]] = False       # UpdateGroupWith: a false clears the group
        else:
            self.reduced[gids[values]] = True         # a true sets it

    def consume(self, values, val_valid, val_off, group_ids, scalar=None):
        """values: bool per row (ignored where the row is null); val_valid / val_off: validity bitmap as elsewhere;
        scalar = (is_valid, value) for a broadcast scalar."""
        gids = np.asarray(group_ids, dtype=np.int64)
        n = len(gids)
        if scalar is not None:
            ok = np.full(n, bool(scalar[0]))
            vals = np.full(n, bool(scalar[1]))
        else:
            ok = _valid_rows(val_valid, val_off, n)
            vals = np.asarray(values, bool)
        np.add.at(self.counts, gids[ok], 1)
        self._update(gids[ok], vals[ok])
        self.no_nulls[gids[~ok]] = False

    def merge(self, other: "HashBoolState", mapping):
        m = np.asarray(mapping, dtype=np.int64)
        np.add.at(self.counts, m, other.counts)
        self._update(m, other.reduced)
        np.logical_and.at(self.no_nulls, m, other.no_nulls)

    def finalize(self):
        """(values, valid) per group."""
        valid = self.counts >= self.min_count
        if not self.skip_nulls:
            decided = ~self.reduced if self.is_all else self.reduced      # BitmapOrNot / BitmapOr with `seen`
            valid = valid & (self.no_nulls | decided)
        return self.reduced.copy(), valid


def delta_binary_packed_encode(values, block_size: int = 128, miniblocks: int = 4) -> bytes:
    """A writer of DELTA_BINARY_PACKED (Encodings.md "Delta encoding"; DeltaBitPackEncoder, parquet/encoder.cc)
    for tests: any block size / miniblock count the format allows, int64 arithmetic modulo 2**64, trailing
    miniblocks without values omitted (their widths stay in the block header, as the format prescribes)."""
    M = (1 << 64) - 1
    v = [int(x) & M for x in values]
    out = bytearray()

    def varint(x):
        while x >= 0x80:
            out.append((x & 0x7F) | 0x80)
            x >>= 7
        out.append(x)

    def zigzag(x):          # x: signed python int in int64 range
        varint(((x << 1) ^ (x >> 63)) & M)

    def signed(u):
        return u - (1 << 64) if u >> 63 else u

    vpm = block_size // miniblocks
    assert block_size % 128 == 0 and block_size % miniblocks == 0 and vpm % 32 == 0
    varint(block_size)
    varint(miniblocks)
    varint(len(v))
    zigzag(signed(v[0]) if v else 0)
    deltas = [signed((v[i] - v[i - 1]) & M) for i in range(1, len(v))]
    for b in range(0, len(deltas), block_size):
        block = deltas[b: b + block_size]
        min_delta = min(block)
        zigzag(min_delta)
        rel = [(d - min_delta) & M for d in block]
        widths = []
        for m in range(miniblocks):
            part = rel[m * vpm: (m + 1) * vpm]
            widths.append(max(part).bit_length() if part else 0)
        out.extend(bytes(widths))
        for m in range(miniblocks):
            part = rel[m * vpm: (m + 1) * vpm]
            if not part:
                break
            part = part + [0] * (vpm - len(part))
            acc = 0
            for k, x in enumerate(part):
                acc |= x << (k * widths[m])
            out.extend(acc.to_bytes(vpm * widths[m] // 8, "little"))
    return bytes(out)


def delta_binary_packed_decode(data: bytes):
    """DeltaBitPackDecoder (parquet/decoder.cc: InitHeader, InitBlock, InitMiniBlock, GetInternal) restated
    value by value: returns (int64 values with wrap-around, bytes consumed)."""
    M = (1 << 64) - 1
    pos = 0

    def varint():
        nonlocal pos
        x, shift = 0, 0
        while True:
            c = data[pos]
            pos += 1
            x |= (c & 0x7F) << shift
            if not c & 0x80:
                return x
            shift += 7

    def zigzag():
        u = varint()
        return (u >> 1) ^ -(u & 1)

    block_size, miniblocks, total = varint(), varint(), varint()
    last = zigzag() & M
    vpm = block_size // miniblocks
    out = [last] if total else []
    while len(out) < total:
        min_delta = zigzag()
        widths = data[pos: pos + miniblocks]
        pos += miniblocks
        for m in range(miniblocks):
            if len(out) >= total:
                break
            w = widths[m]
            nbytes = vpm * w // 8
            acc = int.from_bytes(data[pos: pos + nbytes], "little")
            pos += nbytes
            for k in range(vpm):
                if len(out) >= total:
                    break
                last = (last + min_delta + ((acc >> (k * w)) & ((1 << w) - 1))) & M
                out.append(last)
    return np.array(out, dtype=np.uint64).astype(np.int64), pos


def delta_byte_array_encode(values, block_size: int = 128, miniblocks: int = 4) -> bytes:
    """A writer of DELTA_BYTE_ARRAY (Encodings.md "Delta Strings"; DeltaByteArrayEncoder, parquet/encoder.cc): prefix
    lengths against the previous value (DELTA_BINARY_PACKED), then the suffixes as DELTA_LENGTH_BYTE_ARRAY (their
    lengths DELTA_BINARY_PACKED, then the bytes).  Test infrastructure: block shapes and prefixes the reference's writer
    never produces; the decoder below is what is pinned to the reference."""
    prefixes, suffixes, last = [], [], b""
    for v in values:
        k = 0
        m = min(len(v), len(last))
        while k < m and v[k] == last[k]:
            k += 1
        prefixes.append(k)
        suffixes.append(v[k:])
        last = v
    return (delta_binary_packed_encode(np.asarray(prefixes, dtype=np.int64), block_size, miniblocks) +
            delta_binary_packed_encode(np.asarray([len(x) for x in suffixes], dtype=np.int64), block_size, miniblocks) + b"".join(suffixes))


def delta_byte_array_decode(data: bytes):
    """DeltaByteArrayDecoderImpl (parquet/decoder.cc:1974-2204) restated value by value: SetData (:1988-2018) decodes ALL
    prefix lengths, hands the rest of the page to a DeltaLengthByteArrayDecoder and clears last_value_; GetInternal
    (:2074-2133) / BuildBufferInternal (:2037-2072) build value i from the first prefix[i] bytes of value i - 1 and
    suffix i — "negative prefix length" (:2097) and "prefix length too large" (:2040) are the decoder's errors.
    Returns the list of values (bytes)."""
    prefix, used = delta_binary_packed_decode(data)
    rest = data[used:]
    slen, sused = delta_binary_packed_decode(rest)
    if len(slen) != len(prefix):
        raise ValueError("DELTA_BYTE_ARRAY: prefix and suffix counts differ")
    pos, last, out = sused, b"", []
    for p, n in zip(prefix.tolist(), slen.tolist()):
        if p < 0:
            raise ValueError("negative prefix length in DELTA_BYTE_ARRAY")
        if p > len(last):
            raise ValueError("prefix length too large in DELTA_BYTE_ARRAY")
        if n < 0 or pos + n > len(rest):
            raise ValueError("DELTA_BYTE_ARRAY: suffix runs past the page")
        last = last[:p] + rest[pos: pos + n]
        pos += n
        out.append(last)
    return out


def hash_sum_float_row_order(values, valid, gids, num_groups, sums=None, counts=None, null_seen=None):
    """GroupedReducingAggregator<FloatType / DoubleType, GroupedSumImpl>::Consume (hash_aggregate_numeric.cc:70-83) restated:
    VisitGroupedValues walks the batch in ROW order, a valid value goes into its group's double accumulator with
    Reduce = double(u) + double(v) (:196-206; the accumulator of float32 values is double too, FindAccumulatorType), a null
    clears the group's no_nulls flag.  State arrays are continued when given (the next batch).  Pinned against
    Table.group_by(..., use_threads=False) in tests (one thread = one state = this order)."""
    sums = np.zeros(num_groups, dtype=np.float64) if sums is None else sums
    counts = np.zeros(num_groups, dtype=np.int64) if counts is None else counts
    null_seen = np.zeros(num_groups, dtype=bool) if null_seen is None else null_seen
    v64 = np.asarray(values, dtype=np.float64)
    with np.errstate(all="ignore"):
        for i in range(len(gids)):
            g = int(gids[i])
            if valid is None or valid[i]:
                sums[g] = sums[g] + v64[i]
                counts[g] += 1
            else:
                null_seen[g] = True
    return sums, counts, null_seen


def hash_product_row_order(values, valid, gids, num_groups, products=None, counts=None, null_seen=None):
    """GroupedProductImpl (kernels/hash_aggregate_numeric.cc:311-347) restated: as the grouped sum, with Reduce =
    MultiplyTraits<AccType>::Multiply — the accumulator of an integer column is int64 / uint64 and the product wraps in the
    UNSIGNED type (base_arithmetic_internal.h:303-325: `to_unsigned(u) * to_unsigned(v)`), of a float column double; the state
    starts at 1 (`NullValue` = MultiplyTraits::one).  Row order as VisitGroupedValues walks the batch.  Returns the products as
    uint64 bit patterns (integers) or float64."""
    is_float = np.asarray(values).dtype.kind == "f"
    if products is None:
        products = np.ones(num_groups, dtype=np.float64 if is_float else np.uint64)
    counts = np.zeros(num_groups, dtype=np.int64) if counts is None else counts
    null_seen = np.zeros(num_groups, dtype=bool) if null_seen is None else null_seen
    if is_float:
        v = np.asarray(values, dtype=np.float64)
    else:
        v = np.asarray(values).astype(np.int64).view(np.uint64) if np.asarray(values).dtype.kind == "i" else np.asarray(values).astype(np.uint64)
    with np.errstate(all="ignore"):
        for i in range(len(gids)):
            g = int(gids[i])
            if valid is None or valid[i]:
                products[g] = products[g] * v[i]
                counts[g] += 1
            else:
                null_seen[g] = True
    return products, counts, null_seen


def _neumaier_sum(terms):
    """arrow::internal::NeumaierSum (util/math_internal.h) of a few doubles."""
    total, comp = 0.0, 0.0
    for x in terms:
        t = total + x
        comp += (total - t) + x if abs(total) >= abs(x) else (x - t) + total
        total = t
    return total + comp


def moments_merge(level, a, b):
    """Moments::Merge (kernels/aggregate_var_std_internal.h:116-152); a, b = (count, mean, m2, m3, m4)."""
    if a[0] == 0:
        return b
    if b[0] == 0:
        return a
    na, nb = a[0], b[0]
    n = na + nb
    mean = (a[1] * na + b[1] * nb) / n
    m2 = _neumaier_sum([a[2], b[2], na * (a[1] - mean) * (a[1] - mean), nb * (b[1] - mean) * (b[1] - mean)])
    m3 = m4 = 0.0
    if level >= 3:
        delta = b[1] - a[1]
        delta2 = delta * delta
        m3 = _neumaier_sum([a[3], b[3], delta2 * delta * na * nb * (na - nb) / (n * n), 3 * delta * (na * b[2] - nb * a[2]) / n])
        if level >= 4:
            m4 = _neumaier_sum([a[4], b[4], (delta2 * delta2) * na * nb * (na * na - na * nb + nb * nb) / (n * n * n),
                                6 * delta2 * (na * na * b[2] + nb * nb * a[2]) / (n * n), 4 * delta * (na * b[3] - nb * a[3]) / n])
    return (n, mean, m2, m3, m4)


def grouped_moments(values, valid, gids, num_groups, level=4, state=None):
    """GroupedStatisticImpl::ConsumeGeneric (kernels/hash_aggregate_numeric.cc:555-615) for ONE batch, merged into `state`
    (:700-745): sums in the SumType in row order, mean = ToDouble(sum) / count, then the sums of (ToDouble(x) - mean)^k in row
    order; the batch's moments are merged into the running ones group by group (Moments::Merge).  `state` / the result:
    (list of (count, mean, m2, m3, m4) per group, null_seen).  (Integers of <= 4 bytes take ConsumeIntegral in the
    reference for level 2 — exact integer sums, the same value up to rounding; this restates the generic path for all.)"""
    v = np.asarray(values)
    is_float = v.dtype.kind == "f"
    sums = [0.0 if is_float else 0] * num_groups
    counts = [0] * num_groups
    moments, null_seen = state if state is not None else ([(0, 0.0, 0.0, 0.0, 0.0)] * num_groups, np.zeros(num_groups, dtype=bool))
    moments = list(moments)
    for i in range(len(gids)):
        g = int(gids[i])
        if valid is None or valid[i]:
            sums[g] = sums[g] + (float(v[i]) if is_float else int(v[i]))
            counts[g] += 1
        else:
            null_seen[g] = True
    means = [float(sums[g]) / counts[g] if counts[g] else 0.0 for g in range(num_groups)]
    m = [[0.0, 0.0, 0.0] for _ in range(num_groups)]
    for i in range(len(gids)):
        if valid is not None and not valid[i]:
            continue
        g = int(gids[i])
        d = float(v[i]) - means[g]
        d2 = d * d
        if level >= 4:
            m[g][2] += d2 * d2
        if level >= 3:
            m[g][1] += d2 * d
        m[g][0] += d2
    for g in range(num_groups):
        moments[g] = moments_merge(level, moments[g], (counts[g], means[g], m[g][0], m[g][1], m[g][2]))
    return moments, null_seen


def moments_statistic(moment, stat, ddof=0, biased=True):
    """Moments::Variance / Stddev / Skew / Kurtosis (aggregate_var_std_internal.h:83-114) where GroupedStatisticImpl::Finalize
    (:747-775) computes one, else None.  stat: 0 variance, 1 stddev, 2 skew, 3 kurtosis."""
    count, _, m2, m3, m4 = moment
    if not (count > ddof and (stat != 2 or biased or count > 2) and (stat != 3 or biased or count > 3)):
        return None
    with np.errstate(all="ignore"):
        c = np.float64(count)
        m2, m3, m4 = np.float64(m2), np.float64(m3), np.float64(m4)
        if stat in (0, 1):
            var = m2 / np.float64(count - ddof)
            return float(var if stat == 0 else np.sqrt(var))
        if stat == 2:
            if biased:
                return float(np.sqrt(c) * m3 / np.sqrt(m2 * m2 * m2))
            m2_avg = m2 / c
            return float(np.sqrt(c * (c - 1)) / (c - 2) * (m3 / c) / np.sqrt(m2_avg * m2_avg * m2_avg))
        if biased:
            return float(c * m4 / (m2 * m2) - 3)
        m2_avg = m2 / c
        return float(1.0 / ((c - 2) * (c - 3)) * (((c * c) - 1.0) * (m4 / c) / (m2_avg * m2_avg) - 3 * ((c - 1) * (c - 1))))


def group_edge_rows(gids, valid, num_groups, last=False):
    """What GroupedFirstLastImpl / GroupedOneImpl keep per group (kernels/hash_aggregate.cc:775-808, :1575-1590), as a ROW:
    `firsts[g]` is set by the first NON-NULL value of group g that VisitGroupedValues meets, `lasts[g]` by every non-null
    value (so the last one stays).  Returns (rows, has_row): the row of that value, 0 and False where the group has none."""
    rows = np.zeros(num_groups, dtype=np.uint32)
    has = np.zeros(num_groups, dtype=bool)
    for i in range(len(gids)):
        if valid is not None and not valid[i]:
            continue
        g = int(gids[i])
        if last or not has[g]:
            rows[g] = i
            has[g] = True
    return rows, has
