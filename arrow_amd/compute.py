"""Host-side mirror of the arrow::compute API surface for the hot path, over device arrays.

Same names, argument meaning and error behaviour as the reference:
  FunctionRegistry / Function / Kernel        cpp/src/arrow/compute/registry.h:46, function.h:146-378,
                                              kernel.h:556-769
  CallFunction                                cpp/src/arrow/compute/exec.cc:1362-1390
  Filter / Take / SortIndices / Cast wrappers cpp/src/arrow/compute/api_vector.cc:334-423, cast.cc:239
  FilterOptions / TakeOptions / ...           cpp/src/arrow/compute/api_vector.h:37-104,
                                              api_aggregate.h:48-59
Every kernel `exec` here only allocates outputs, applies the reference's host-side decisions
(output validity allocation, null_count bookkeeping, shape dispatch) and calls the C ABI of
libarrow_amd.so (include/arrow_amd.h).  There is no CPU compute path.
"""
from __future__ import annotations

import builtins
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib, tracing
from ._lib import (ArrowIndexError, ArrowInvalid, ArrowNotImplementedError, check)  # noqa: F401
from .array import (Array, DataType, RunEndEncoded, Scalar, alloc, bitmap_nbytes, bool_, current_stream, float32,
                    float64, int8, int16, int32, int64, kUnknownNullCount, uint8, uint16, uint32, uint64,
                    INDEX_TYPE_ID, is_base_binary)

# --------------------------------------------------------------------------- options
class FunctionOptions:
    pass


class FilterOptions(FunctionOptions):
    """api_vector.h:37-52.  null_selection_behavior: 'drop' | 'emit_null'."""

    def __init__(self, null_selection_behavior: str = "drop"):
        if null_selection_behavior not in ("drop", "emit_null"):
            raise ArrowInvalid(f"invalid null_selection_behavior {null_selection_behavior!r}")
        self.null_selection_behavior = null_selection_behavior

    @property
    def code(self) -> int:
        return _lib.FILTER_EMIT_NULL if self.null_selection_behavior == "emit_null" else _lib.FILTER_DROP


class TakeOptions(FunctionOptions):
    """api_vector.h:54-63."""

    def __init__(self, boundscheck: bool = True):
        self.boundscheck = bool(boundscheck)


class CastOptions(FunctionOptions):
    def __init__(self, to_type: DataType, allow_float_truncate: bool = False, allow_int_overflow: bool = False):
        self.to_type = to_type
        self.allow_float_truncate = allow_float_truncate
        self.allow_int_overflow = allow_int_overflow        # CastOptions::Unsafe sets it (cast.h:60-75)


class ArraySortOptions(FunctionOptions):
    """api_vector.h:93-104."""

    def __init__(self, order: str = "ascending", null_placement: str = "at_end"):
        if order not in ("ascending", "descending"):
            raise ArrowInvalid(f"invalid sort order {order!r}")
        if null_placement not in ("at_start", "at_end"):
            raise ArrowInvalid(f"invalid null placement {null_placement!r}")
        self.order = order
        self.null_placement = null_placement


class SortOptions(FunctionOptions):
    """api_vector.h:106-: sort_keys = [(name, order)]; for a plain Array only the order is used."""

    def __init__(self, sort_keys=(("", "ascending"),), null_placement: str = "at_end"):
        self.sort_keys = list(sort_keys)
        self.null_placement = null_placement


class ScalarAggregateOptions(FunctionOptions):
    """api_aggregate.h:48-59."""

    def __init__(self, skip_nulls: bool = True, min_count: int = 1):
        self.skip_nulls = bool(skip_nulls)
        self.min_count = int(min_count)


# --------------------------------------------------------------------------- scratch space
_tls = threading.local()


def _workspace(device: torch.device, nbytes: int, slot: str = "ws") -> torch.Tensor:
    """Per-thread, per-device scratch buffer (the C ABI never allocates)."""
    cache = getattr(_tls, "cache", None)
    if cache is None:
        cache = _tls.cache = {}
    key = (slot, str(device))
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = cache[key] = alloc(max(nbytes, 1 << 16), device)
    return buf


def _lib_and_stream(device: torch.device):
    return _lib.get_lib(), current_stream(device)


def copy_buffer(src: torch.Tensor, dst: torch.Tensor | None = None) -> torch.Tensor:
    """Device-to-device copy of a uint8 buffer with the HIP kernel (arx_buffer_copy) — the kROCM -> kROCM leg of
    MemoryManager::CopyBufferTo (cpp/src/arrow/device.h:214-222).  Returns `dst` (allocated when None)."""
    lib, stream = _lib_and_stream(src.device)
    if dst is None:
        dst = alloc(src.numel(), src.device)
    if dst.numel() < src.numel():
        raise ValueError("copy_buffer: destination smaller than source")
    check(lib.arx_buffer_copy(src.data_ptr(), dst.data_ptr(), src.numel(), stream))
    return dst


# --------------------------------------------------------------------------- kernels (exec)
def _propagate_validity(args, length: int, device):
    """NullHandling::INTERSECTION (PropagateNullsSpans, exec.cc:1222-1281) on the device.
    Returns (validity buffer or None, null_count)."""
    arrs = [a for a in args if isinstance(a, Array) and a.may_have_nulls()]
    for a in args:
        if isinstance(a, Scalar) and not a.is_valid:
            buf = alloc(bitmap_nbytes(length), device, zero=True)  # all null
            return buf, length
    if not arrs:
        return None, 0
    lib, stream = _lib_and_stream(device)
    if len(arrs) == 1 and arrs[0].offset == 0:
        return arrs[0].validity, arrs[0].null_count  # zero-copy share
    out = alloc(bitmap_nbytes(length), device)
    first = arrs[0]
    if len(arrs) == 1:
        check(lib.arx_bitmap_copy(first.validity.data_ptr(), first.offset, length, out.data_ptr(), stream))
        return out, first.null_count
    check(lib.arx_bitmap_and(first.validity.data_ptr(), first.offset, arrs[1].validity.data_ptr(),
                             arrs[1].offset, length, out.data_ptr(), stream))
    for a in arrs[2:]:
        check(lib.arx_bitmap_and(out.data_ptr(), 0, a.validity.data_ptr(), a.offset, length,
                                 out.data_ptr(), stream))
    return out, kUnknownNullCount


def _exec_array_filter(args, options):
    """PrimitiveFilterExec (vector_selection_filter_internal.cc:445-510)."""
    values, mask = args
    options = options or FilterOptions()
    if mask.type != bool_:
        raise ArrowNotImplementedError("filter: the selection must be a boolean array")
    if values.type == bool_:
        return _exec_boolean_filter(args, options)
    if values.length != mask.length:  # ExecSpanIterator::Init, exec.cc:349-355
        raise ArrowInvalid("Array arguments must all be the same length")
    dev = values.device
    lib, stream = _lib_and_stream(dev)
    n = mask.length
    ws_bytes = lib.arx_filter_workspace_bytes(n)
    ws = _workspace(dev, ws_bytes)
    mspan, vspan = mask.span(), values.span()
    out_len = C.c_int64(0)
    check(lib.arx_filter_count(C.byref(mspan), options.code, ws.data_ptr(), ws.numel(),
                               C.byref(out_len), stream))
    s = out_len.value
    filter_null_count_is_zero = mask.null_count == 0
    drop = options.code == _lib.FILTER_DROP
    null_count = 0 if (values.null_count == 0 and (drop or filter_null_count_is_zero)) else kUnknownNullCount
    allocate_validity = values.null_count != 0 or not filter_null_count_is_zero
    w = values.type.byte_width
    out_data = alloc(s * w, dev)
    out_valid = alloc(bitmap_nbytes(s), dev) if allocate_validity else None
    with tracing.span("arx_filter_exec"):
        check(lib.arx_filter_exec(C.byref(vspan), w, C.byref(mspan), options.code, ws.data_ptr(), s,
                                  out_data.data_ptr(),
                                  None if out_valid is None else out_valid.data_ptr(), stream))
    return Array(values.type, s, [out_valid, out_data], null_count, 0)


def get_take_indices(mask: Array, null_selection_behavior: str = "drop") -> Array:
    """GetTakeIndices (vector_selection_take_internal.cc:258-305): uint16 for <= 65535 rows."""
    options = FilterOptions(null_selection_behavior)
    if mask.type != bool_:
        raise ArrowNotImplementedError("GetTakeIndices needs a boolean array")
    dev = mask.device
    lib, stream = _lib_and_stream(dev)
    n = mask.length
    if n > 0xFFFFFFFF:
        raise ArrowNotImplementedError(
            "Filter length exceeds UINT32_MAX, consider a different strategy for selecting elements")
    ws = _workspace(dev, lib.arx_filter_workspace_bytes(n))
    mspan = mask.span()
    out_len = C.c_int64(0)
    check(lib.arx_filter_count(C.byref(mspan), options.code, ws.data_ptr(), ws.numel(),
                               C.byref(out_len), stream))
    s = out_len.value
    itype = uint16 if n <= 65535 else uint32
    emit = options.code == _lib.FILTER_EMIT_NULL and mask.may_have_nulls()
    out = alloc(s * itype.byte_width, dev)
    out_valid = alloc(bitmap_nbytes(s), dev) if emit else None
    with tracing.span("arx_mask_to_indices"):
        check(lib.arx_mask_to_indices(C.byref(mspan), options.code, ws.data_ptr(), s, itype.byte_width,
                                      out.data_ptr(),
                                      None if out_valid is None else out_valid.data_ptr(), stream))
    return Array(itype, s, [out_valid, out], kUnknownNullCount if emit else 0, 0)


class _LazyCount:
    """null_count = length - valid_count, resolved on first read (take sets it exactly, :377)."""

    def __init__(self, length, counter):
        self.length, self.counter = length, counter

    def __call__(self):
        return int(self.length - int(self.counter.cpu().view(torch.int64)[0]))


def _exec_array_take(args, options):
    """FixedWidthTakeExec (vector_selection_take_internal.cc:405-468)."""
    values, indices = args
    options = options or TakeOptions()
    if indices.type.name not in INDEX_TYPE_ID:
        raise ArrowNotImplementedError(f"take: unsupported index type {indices.type.name}")
    dev = values.device
    lib, stream = _lib_and_stream(dev)
    tid = INDEX_TYPE_ID[indices.type.name]
    ispan, vspan = indices.span(), values.span()
    if values.type == bool_:
        return _take_boolean(values, indices, options, lib, stream, tid, ispan, vspan)
    if options.boundscheck:
        ws = _workspace(dev, lib.arx_take_workspace_bytes(), "take")
        check(lib.arx_check_index_bounds(C.byref(ispan), tid, values.length, ws.data_ptr(), ws.numel(),
                                         stream))
    m = indices.length
    w = values.type.byte_width
    allocate_validity = values.may_have_nulls() or indices.may_have_nulls()
    out_data = alloc(m * w, dev)
    out_valid = alloc(bitmap_nbytes(m), dev) if allocate_validity else None
    counter = None
    if allocate_validity:
        counter = torch.zeros(8, dtype=torch.uint8, device=dev)
    with tracing.span("arx_take"):
        check(lib.arx_take(C.byref(vspan), w, C.byref(ispan), tid, out_data.data_ptr(),
                           None if out_valid is None else out_valid.data_ptr(),
                           None if counter is None else counter.data_ptr(), stream))
    out = Array(values.type, m, [out_valid, out_data], 0, 0)
    if allocate_validity:
        out.set_lazy_null_count(_LazyCount(m, counter))
    return out


def _take_boolean(values, indices, options, lib, stream, tid, ispan, vspan):
    """Take on bit-packed boolean values (the 1-bit Gather of gather_internal.h)."""
    dev = values.device
    if options.boundscheck:
        ws = _workspace(dev, lib.arx_take_workspace_bytes(), "take")
        check(lib.arx_check_index_bounds(C.byref(ispan), tid, values.length, ws.data_ptr(), ws.numel(), stream))
    m = indices.length
    allocate_validity = values.may_have_nulls() or indices.may_have_nulls()
    out_bits = alloc(bitmap_nbytes(m), dev, zero=True)
    out_valid = alloc(bitmap_nbytes(m), dev, zero=True) if allocate_validity else None
    counter = torch.zeros(8, dtype=torch.uint8, device=dev) if allocate_validity else None
    check(lib.arx_take_bits(C.byref(vspan), C.byref(ispan), tid, out_bits.data_ptr(),
                            None if out_valid is None else out_valid.data_ptr(),
                            None if counter is None else counter.data_ptr(), stream))
    out = Array(bool_, m, [out_valid, out_bits], 0, 0)
    if allocate_validity:
        out.set_lazy_null_count(_LazyCount(m, counter))
    return out


def _exec_boolean_filter(args, options):
    """Filter on boolean values = take of GetTakeIndices(mask), like the var-width types."""
    values, mask = args
    options = options or FilterOptions()
    if values.length != mask.length:
        raise ArrowInvalid("Array arguments must all be the same length")
    indices = get_take_indices(mask, options.null_selection_behavior)
    return _exec_array_take([values, indices], TakeOptions(boundscheck=False))


def _exec_binary_take(args, options):
    """TakeExec for base binary (vector_selection_take_internal.cc, VarBinary take): offsets,
    validity and the byte total first (the reference grows a builder), then the bytes."""
    values, indices = args
    options = options or TakeOptions()
    if indices.type.name not in INDEX_TYPE_ID:
        raise ArrowNotImplementedError(f"take: unsupported index type {indices.type.name}")
    dev = values.device
    lib, stream = _lib_and_stream(dev)
    tid = INDEX_TYPE_ID[indices.type.name]
    ispan, vspan = indices.span(), values.binary_span()
    if options.boundscheck:
        ws = _workspace(dev, lib.arx_take_workspace_bytes(), "take")
        check(lib.arx_check_index_bounds(C.byref(ispan), tid, values.length, ws.data_ptr(), ws.numel(),
                                         stream))
    m = indices.length
    allocate_validity = values.may_have_nulls() or indices.may_have_nulls()
    out_offsets = alloc((m + 1) * 4, dev)
    out_valid = alloc(bitmap_nbytes(m), dev) if allocate_validity else None
    counter = torch.zeros(8, dtype=torch.uint8, device=dev) if allocate_validity else None
    ws = _workspace(dev, lib.arx_binary_take_workspace_bytes(m), "binary_take")
    total = C.c_int64(0)
    with tracing.span("arx_binary_take_offsets"):
        check(lib.arx_binary_take_offsets(C.byref(vspan), C.byref(ispan), tid, ws.data_ptr(), ws.numel(),
                                          out_offsets.data_ptr(),
                                          None if out_valid is None else out_valid.data_ptr(),
                                          None if counter is None else counter.data_ptr(),
                                          C.byref(total), stream))
    out_data = alloc(total.value, dev)
    with tracing.span("arx_binary_take_data"):
        check(lib.arx_binary_take_data(C.byref(vspan), m, ws.data_ptr(), ws.numel(), out_offsets.data_ptr(),
                                       total.value, out_data.data_ptr(), stream))
    out = Array(values.type, m, [out_valid, out_offsets, out_data], 0, 0)
    if allocate_validity:
        out.set_lazy_null_count(_LazyCount(m, counter))
    return out


def _exec_binary_filter(args, options):
    """BinaryFilterImpl (vector_selection_filter_internal.cc:517-800) as take(GetTakeIndices(mask)):
    the same output by construction (GetTakeIndices emits exactly the slots the filter keeps, with
    nulls for EMIT_NULL)."""
    values, mask = args
    options = options or FilterOptions()
    if mask.type != bool_:
        raise ArrowNotImplementedError("filter: the selection must be a boolean array")
    if values.length != mask.length:
        raise ArrowInvalid("Array arguments must all be the same length")
    indices = get_take_indices(mask, options.null_selection_behavior)
    return _exec_binary_take([values, indices], TakeOptions(boundscheck=False))


def _exec_cast_f64_f32(args, options):
    """CastFloatingToFloating (scalar_cast_numeric.cc:56-60) under ScalarExecutor."""
    (arr,) = args
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    out = alloc(arr.length * 4, dev)
    with tracing.span("arx_cast_f64_f32"):
        check(lib.arx_cast_f64_f32(arr.values_ptr(), arr.length, out.data_ptr(), stream))
    validity, nc = _propagate_validity([arr], arr.length, dev)
    return Array(float32, arr.length, [validity, out], nc, 0)


def _exec_cast_i64_i32(args, options):
    """CastIntegerToInteger (scalar_cast_numeric.cc:46-54): IntegersCanFit unless allow_int_overflow."""
    (arr,) = args
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    out = alloc(n * 4, dev)
    ws = _workspace(dev, 64, "cast")
    sp = arr.span()
    check(lib.arx_cast_i64_i32(C.byref(sp), int(bool(getattr(options, "allow_int_overflow", False))), ws.data_ptr(),
                               ws.numel(), out.data_ptr(), stream))
    validity, nc = _propagate_validity([arr], n, dev)
    return Array(int32, n, [validity, out], nc, 0)


def _exec_cast_i64_f64(args, options):
    """CastIntegerToFloating (scalar_cast_numeric.cc:270-279): exactness check unless allow_float_truncate."""
    (arr,) = args
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    out = alloc(n * 8, dev)
    ws = _workspace(dev, 64, "cast")
    sp = arr.span()
    check(lib.arx_cast_i64_f64(C.byref(sp), int(bool(getattr(options, "allow_float_truncate", False))), ws.data_ptr(),
                               ws.numel(), out.data_ptr(), stream))
    validity, nc = _propagate_validity([arr], n, dev)
    return Array(float64, n, [validity, out], nc, 0)


# ARX_NUM_* of include/arrow_amd.h
_NUM_TYPE_ID = {"int8": 0, "uint8": 1, "int16": 2, "uint16": 3, "int32": 4, "uint32": 5, "int64": 6, "uint64": 7,
                "float": 8, "double": 9}
_NUMERIC = lambda t: t.name in _NUM_TYPE_ID  # noqa: E731


def _make_exec_cast_numeric(to_type: DataType):
    def _exec(args, options):
        """Any numeric pair (scalar_cast_numeric.cc:46-60, 190-207, 270-279): static_cast on every slot + the mode's
        check on the valid ones (arx_cast_numeric)."""
        (arr,) = args
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        n = arr.length
        out = alloc(n * to_type.byte_width, dev)
        ws = _workspace(dev, 64, "cast")
        sp = arr.span()
        check(lib.arx_cast_numeric(C.byref(sp), _NUM_TYPE_ID[arr.type.name], _NUM_TYPE_ID[to_type.name],
                                   int(bool(getattr(options, "allow_int_overflow", False))),
                                   int(bool(getattr(options, "allow_float_truncate", False))),
                                   ws.data_ptr(), ws.numel(), out.data_ptr(), stream))
        validity, nc = _propagate_validity([arr], n, dev)
        return Array(to_type, n, [validity, out], nc, 0)
    return _exec


def _exec_cast_i32_i64(args, options):
    (arr,) = args
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    out = alloc(n * 8, dev)
    check(lib.arx_cast_i32_i64(arr.values_ptr(), n, out.data_ptr(), stream))
    validity, nc = _propagate_validity([arr], n, dev)
    return Array(int64, n, [validity, out], nc, 0)


def _scalar_value(x):
    return x.value if isinstance(x, Scalar) else x


def _exec_greater(args, options):
    """CompareKernel<..., Greater>::Exec (scalar_compare.cc:259-298)."""
    left, right = args
    arr = left if isinstance(left, Array) else right
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    out = alloc(bitmap_nbytes(n), dev, zero=True)
    t = arr.type
    if isinstance(left, Array) and isinstance(right, Array):
        if left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        fn = lib.arx_greater_f64 if t == float64 else lib.arx_greater_i64
        with tracing.span("arx_greater"):
            check(fn(left.values_ptr(), right.values_ptr(), n, out.data_ptr(), stream))
    elif isinstance(left, Array):
        if t == float64:
            check(lib.arx_greater_f64_array_scalar(left.values_ptr(), float(_scalar_value(right) or 0.0), n,
                                                   out.data_ptr(), stream))
        else:
            check(lib.arx_greater_i64_array_scalar(left.values_ptr(), int(_scalar_value(right) or 0), n,
                                                   out.data_ptr(), stream))
    else:
        if t == float64:
            check(lib.arx_greater_f64_scalar_array(float(_scalar_value(left) or 0.0), right.values_ptr(), n,
                                                   out.data_ptr(), stream))
        else:
            check(lib.arx_greater_i64_scalar_array(int(_scalar_value(left) or 0), right.values_ptr(), n,
                                                   out.data_ptr(), stream))
    validity, nc = _propagate_validity([left, right], n, dev)
    return Array(bool_, n, [validity, out], nc, 0)


_ARITH_CODE = {"add": 0, "subtract": 1, "multiply": 2}


def _exec_arith(op_name, checked):
    code = _ARITH_CODE[op_name]

    def run(args, options):
        """ScalarBinary<..., Add|Subtract|Multiply> / ScalarBinaryNotNull<..., *Checked>
        (codegen_internal.h:814-; base_arithmetic_internal.h:45-150,290-364)."""
        left, right = args
        arr = left if isinstance(left, Array) else right
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        n = arr.length
        if isinstance(left, Array) and isinstance(right, Array) and left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        is_f = arr.type == float64
        conv = float if is_f else int
        lp = left.values_ptr() if isinstance(left, Array) else None
        rp = right.values_ptr() if isinstance(right, Array) else None
        ls = conv(_scalar_value(left) or 0) if lp is None else conv(0)
        rs = conv(_scalar_value(right) or 0) if rp is None else conv(0)
        out = alloc(n * 8, dev)
        validity, nc = _propagate_validity([left, right], n, dev)
        null_scalar = any(isinstance(a, Scalar) and not a.is_valid for a in (left, right))
        if is_f:
            check(lib.arx_arith_f64(code, lp, ls, rp, rs, n, out.data_ptr(), stream))
        elif not checked or null_scalar:      # a null scalar makes every slot null: nothing to check
            check(lib.arx_arith_i64(code, lp, ls, rp, rs, n, out.data_ptr(), stream))
        else:
            flag = torch.zeros(1, dtype=torch.int32, device=dev)

            def vptr(a):
                return (a.validity.data_ptr(), a.offset) if isinstance(a, Array) and a.may_have_nulls() else (None, 0)
            (lvp, lvo), (rvp, rvo) = vptr(left), vptr(right)
            check(lib.arx_arith_checked_i64(code, lp, ls, lvp, lvo, rp, rs, rvp, rvo, n, out.data_ptr(),
                                            flag.data_ptr(), stream))
            if int(flag.cpu()[0]) != 0:
                raise ArrowInvalid("overflow")   # AddChecked::Call, base_arithmetic_internal.h:77
        return Array(arr.type, n, [validity, out], nc, 0)
    return run


def _exec_divide(checked: bool):
    def run(args, options):
        """ScalarBinaryNotNull<..., Divide | DivideChecked> (base_arithmetic_internal.h:366-424): only slots where both
        operands are valid are visited; the last failing slot names the error."""
        left, right = args
        arr = left if isinstance(left, Array) else right
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        n = arr.length
        if isinstance(left, Array) and isinstance(right, Array) and left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        is_f = arr.type == float64
        conv = float if is_f else int
        lp = left.values_ptr() if isinstance(left, Array) else None
        rp = right.values_ptr() if isinstance(right, Array) else None
        null_scalar = any(isinstance(a, Scalar) and not a.is_valid for a in (left, right))
        ls = conv(_scalar_value(left) or 0) if lp is None else conv(0)
        rs = conv(_scalar_value(right) or 0) if rp is None else conv(0)
        if null_scalar:           # every slot is null: nothing is visited (keep the kernel away from a 0 divisor)
            ls, rs = (ls, conv(1)) if rp is None else (ls, rs)
        out = alloc(n * 8, dev)
        validity, nc = _propagate_validity([left, right], n, dev)
        errors = torch.zeros(2, dtype=torch.int64, device=dev)

        def vptr(a):
            return (a.validity.data_ptr(), a.offset) if isinstance(a, Array) and a.may_have_nulls() else (None, 0)
        (lvp, lvo), (rvp, rvo) = vptr(left), vptr(right)
        fn = lib.arx_divide_f64 if is_f else lib.arx_divide_i64
        check(fn(lp, ls, lvp, lvo, rp, rs, rvp, rvo, n, 1 if checked else 0, out.data_ptr(), errors.data_ptr(), stream))
        if not null_scalar:
            last_overflow, last_zero = errors.cpu().tolist()
            if last_zero > last_overflow:
                raise ArrowInvalid("divide by zero")
            if last_overflow:
                raise ArrowInvalid("overflow")
        return Array(arr.type, n, [validity, out], nc, 0)
    return run


_CMP_CODE = {"equal": 0, "not_equal": 1, "greater": 2, "greater_equal": 3, "less": 4, "less_equal": 5}


def _exec_compare(op_name):
    code = _CMP_CODE[op_name]

    def run(args, options):
        """CompareKernel<..., Op>::Exec (scalar_compare.cc:259-298) for Op in Equal ... LessEqual."""
        left, right = args
        arr = left if isinstance(left, Array) else right
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        n = arr.length
        if isinstance(left, Array) and isinstance(right, Array) and left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        out = alloc(bitmap_nbytes(n), dev, zero=True)
        is_f = arr.type == float64
        conv = float if is_f else int
        lp = left.values_ptr() if isinstance(left, Array) else None
        rp = right.values_ptr() if isinstance(right, Array) else None
        ls = conv(_scalar_value(left) or 0) if lp is None else conv(0)
        rs = conv(_scalar_value(right) or 0) if rp is None else conv(0)
        fn = lib.arx_compare_f64 if is_f else lib.arx_compare_i64
        with tracing.span("arx_compare"):
            check(fn(code, lp, ls, rp, rs, n, out.data_ptr(), stream))
        validity, nc = _propagate_validity([left, right], n, dev)
        return Array(bool_, n, [validity, out], nc, 0)
    return run


def _numeric_operands(left, right):
    """(array operand, n, left ptr | None, right ptr | None, host scalar holders) for the *_numeric entry points."""
    arr = left if isinstance(left, Array) else right
    n = arr.length
    if isinstance(left, Array) and isinstance(right, Array) and left.length != right.length:
        raise ArrowInvalid("Array arguments must all be the same length")
    dt = arr.type.np_dtype

    def scalar_ptr(x):
        if isinstance(x, Array):
            return None, None
        v = _scalar_value(x)
        holder = np.zeros(1, dtype=dt)
        if v is not None:
            holder[0] = dt.type(v)
        return holder, holder.ctypes.data

    lh, lsp = scalar_ptr(left)
    rh, rsp = scalar_ptr(right)
    lp = left.values_ptr() if isinstance(left, Array) else None
    rp = right.values_ptr() if isinstance(right, Array) else None
    return arr, n, lp, lsp, rp, rsp, (lh, rh)


def _num_type_id(t: DataType) -> int:
    """ARX_NUM_* of a numeric type, or of the physical integer of a temporal one (timestamp / duration / time / date)."""
    if t.name in _NUM_TYPE_ID:
        return _NUM_TYPE_ID[t.name]
    from .array import is_temporal

    if is_temporal(t):
        return _NUM_TYPE_ID["int64" if t.bit_width == 64 else "int32"]
    raise ArrowNotImplementedError(f"no numeric kernel for {t.name}")


def _exec_compare_numeric(op_name):
    code = _CMP_CODE[op_name]

    def run(args, options):
        """CompareKernel<Type, Op>::Exec (scalar_compare.cc:259-298) for every numeric Type (registration :398-446)."""
        left, right = args
        arr, n, lp, lsp, rp, rsp, _keep = _numeric_operands(left, right)
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        out = alloc(bitmap_nbytes(n), dev, zero=True)
        with tracing.span("arx_compare_numeric"):
            check(lib.arx_compare_numeric(code, _num_type_id(arr.type), lp, lsp, rp, rsp, n, out.data_ptr(), stream))
        validity, nc = _propagate_validity([left, right], n, dev)
        return Array(bool_, n, [validity, out], nc, 0)
    return run


def _exec_arith_numeric(op_name, checked):
    code = _ARITH_CODE[op_name]

    def run(args, options):
        """ScalarBinary<T, T, T, Add|Subtract|Multiply> / ScalarBinaryNotNull<..., *Checked> for every numeric T
        (base_arithmetic_internal.h:45-150,290-364)."""
        left, right = args
        arr, n, lp, lsp, rp, rsp, _keep = _numeric_operands(left, right)
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        out = alloc(n * arr.type.byte_width, dev)
        validity, nc = _propagate_validity([left, right], n, dev)
        null_scalar = any(isinstance(a, Scalar) and not a.is_valid for a in (left, right))
        is_f = arr.type.name in ("float", "double")
        do_check = checked and not is_f and not null_scalar      # a null scalar makes every slot null: nothing to check
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if do_check else None

        def vptr(a):
            return (a.validity.data_ptr(), a.offset) if isinstance(a, Array) and a.may_have_nulls() else (None, 0)
        (lvp, lvo), (rvp, rvo) = vptr(left), vptr(right)
        with tracing.span("arx_arith_numeric"):
            check(lib.arx_arith_numeric(code, 1 if do_check else 0, _NUM_TYPE_ID[arr.type.name], lp, lsp, lvp, lvo, rp, rsp,
                                        rvp, rvo, n, out.data_ptr(), None if flag is None else flag.data_ptr(), stream))
        if flag is not None and int(flag.cpu()[0]) != 0:
            raise ArrowInvalid("overflow")   # AddChecked::Call, base_arithmetic_internal.h:77
        return Array(arr.type, n, [validity, out], nc, 0)
    return run


def _exec_divide_numeric(checked: bool):
    def run(args, options):
        """ScalarBinaryNotNull<T, T, T, Divide | DivideChecked> for every numeric T (base_arithmetic_internal.h:366-424,
        DivideWithOverflowGeneric util/int_util_overflow.h:124-138): only slots where both operands are valid are
        visited; the last failing slot names the error."""
        left, right = args
        arr, n, lp, lsp, rp, rsp, _keep = _numeric_operands(left, right)
        dev = arr.device
        lib, stream = _lib_and_stream(dev)
        null_scalar = any(isinstance(a, Scalar) and not a.is_valid for a in (left, right))
        if null_scalar and rp is None:      # every slot is null: nothing is visited (keep the kernel away from a 0 divisor)
            _keep[1][0] = 1
        out = alloc(n * arr.type.byte_width, dev)
        validity, nc = _propagate_validity([left, right], n, dev)
        errors = torch.zeros(2, dtype=torch.int64, device=dev)

        def vptr(a):
            return (a.validity.data_ptr(), a.offset) if isinstance(a, Array) and a.may_have_nulls() else (None, 0)
        (lvp, lvo), (rvp, rvo) = vptr(left), vptr(right)
        with tracing.span("arx_divide_numeric"):
            check(lib.arx_divide_numeric(1 if checked else 0, _NUM_TYPE_ID[arr.type.name], lp, lsp, lvp, lvo, rp, rsp, rvp, rvo, n,
                                         out.data_ptr(), errors.data_ptr(), stream))
        if not null_scalar:
            last_overflow, last_zero = errors.cpu().tolist()
            if last_zero > last_overflow:
                raise ArrowInvalid("divide by zero")
            if last_overflow:
                raise ArrowInvalid("overflow")
        return Array(arr.type, n, [validity, out], nc, 0)
    return run


def _exec_add(args, options):
    """ScalarBinary<..., Add> (codegen_internal.h:814, base_arithmetic_internal.h:45-80)."""
    left, right = args
    if not isinstance(left, Array):      # scalar + array: add commutes
        left, right = right, left
    dev = left.device
    lib, stream = _lib_and_stream(dev)
    n = left.length
    out = alloc(n * 8, dev)
    if isinstance(right, Array):
        if left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        fn = lib.arx_add_i64 if left.type == int64 else lib.arx_add_f64
        check(fn(left.values_ptr(), right.values_ptr(), n, out.data_ptr(), stream))
    elif left.type == int64:
        check(lib.arx_add_i64_array_scalar(left.values_ptr(), int(_scalar_value(right) or 0), n, out.data_ptr(),
                                           stream))
    else:
        check(lib.arx_add_f64_array_scalar(left.values_ptr(), float(_scalar_value(right) or 0.0), n,
                                           out.data_ptr(), stream))
    validity, nc = _propagate_validity([left, right], n, dev)
    return Array(left.type, n, [validity, out], nc, 0)


def _exec_kleene(op_code):
    def run(args, options):
        """KleeneAndOp / KleeneOrOp (scalar_boolean.cc:138-260), array x array."""
        left, right = args
        if not (isinstance(left, Array) and isinstance(right, Array)):
            raise ArrowNotImplementedError("and_kleene / or_kleene with a scalar operand is not on the gfx950 path")
        if left.length != right.length:
            raise ArrowInvalid("Array arguments must all be the same length")
        dev = left.device
        lib, stream = _lib_and_stream(dev)
        n = left.length
        out = alloc(bitmap_nbytes(n), dev, zero=True)
        nulls = left.may_have_nulls() or right.may_have_nulls()
        validity = alloc(bitmap_nbytes(n), dev, zero=True) if nulls else None
        ls, rs = left.span(), right.span()
        check(lib.arx_boolean_kleene(op_code, C.byref(ls), C.byref(rs), out.data_ptr(),
                                     None if validity is None else validity.data_ptr(), stream))
        return Array(bool_, n, [validity, out], kUnknownNullCount if nulls else 0, 0)
    return run


def _exec_invert(args, options):
    """InvertOp (scalar_boolean.cc:39-50): data inverted, validity propagated."""
    (arr,) = args
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    out = alloc(bitmap_nbytes(n), dev, zero=True)
    check(lib.arx_boolean_invert(arr.data.data_ptr(), arr.offset, n, out.data_ptr(), stream))
    validity, nc = _propagate_validity([arr], n, dev)
    return Array(bool_, n, [validity, out], nc, 0)


# ARX_KEY_* of include/arrow_amd.h
_SORT_KEY_TYPE = {"uint64": 0, "int64": 1, "uint32": 2, "int32": 3, "double": 4, "float": 5}


def _exec_array_sort_indices(args, options):
    """ArraySortIndices::Exec (vector_array_sort.cc:524-540): uint64 indices, never null."""
    (arr,) = args
    options = options or ArraySortOptions()
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    ws_bytes = lib.arx_sort_indices_workspace_bytes(n)
    ws = _workspace(dev, ws_bytes + 256, "sort")
    base = ws.data_ptr()
    aligned = (base + 255) & ~255
    out = alloc(n * 8, dev)
    span = arr.span()
    check(lib.arx_sort_indices(C.byref(span), _SORT_KEY_TYPE[arr.type.name],
                                  _lib.SORT_DESCENDING if options.order == "descending" else _lib.SORT_ASCENDING,
                                  _lib.NULLS_AT_START if options.null_placement == "at_start" else _lib.NULLS_AT_END,
                                  aligned, ws.numel() - (aligned - base), out.data_ptr(), stream))
    return Array(uint64, n, [None, out], 0, 0)


def _exec_array_sort_indices_bool(args, options):
    """array_sort_indices(boolean): the reference counts (ArrayCountSorter<BooleanType>, vector_array_sort.cc:360-400) —
    the rows of the falses, of the trues (descending: trues first) and of the nulls at their end, each class in row
    order: three GetTakeIndices over the value bitmap, its inverse and the inverted validity."""
    (arr,) = args
    options = options or ArraySortOptions()
    n = arr.length
    if n == 0:
        return Array(uint64, 0, [None, alloc(8, arr.device)], 0, 0)
    trues = indices_nonzero(arr)
    falses = indices_nonzero(invert(arr))
    parts = [trues, falses] if options.order == "descending" else [falses, trues]
    if arr.validity is not None and arr.null_count != 0:
        nulls = indices_nonzero(invert(Array(bool_, n, [None, arr.validity], 0, arr.offset)))
        parts = [nulls] + parts if options.null_placement == "at_start" else parts + [nulls]
    out = torch.cat([p.data[: p.length * 8] for p in parts])
    assert out.numel() == n * 8
    return Array(uint64, n, [None, out], 0, 0)


# --------------------------------------------------------------------------- registry
class Kernel:
    """arrow::compute::Kernel twin: a signature (tuple of DataType or None = any) + exec."""

    def __init__(self, in_types, exec_fn, out_type=None):
        self.in_types = tuple(in_types)
        self.exec = exec_fn
        self.out_type = out_type

    def matches(self, types) -> bool:
        if len(types) != len(self.in_types):
            return False
        return all(want is None or (callable(want) and want(t)) or want == t
                   for want, t in zip(self.in_types, types))


class Function:
    SCALAR, VECTOR, SCALAR_AGGREGATE, HASH_AGGREGATE, META = range(5)  # function.h:146-168

    def __init__(self, name: str, kind: int, arity: int, default_options=None, meta_impl=None):
        self.name = name
        self.kind = kind
        self.arity = arity
        self.default_options = default_options
        self.kernels: list[Kernel] = []
        self._meta_impl = meta_impl

    def add_kernel(self, kernel: Kernel) -> None:
        if len(kernel.in_types) != self.arity:  # function.cc:448-455
            raise ArrowInvalid(f"Added kernel accepts {len(kernel.in_types)} arguments but the "
                               f"function {self.name} accepts {self.arity}")
        self.kernels.append(kernel)

    @property
    def num_kernels(self) -> int:
        return len(self.kernels)

    def dispatch_exact(self, types) -> Kernel:
        """DispatchExactImpl (function.cc:122-157): the LAST matching kernel wins."""
        for k in reversed(self.kernels):
            if k.matches(types):
                return k
        names = ", ".join(t.name for t in types)
        raise ArrowNotImplementedError(
            f"Function '{self.name}' has no kernel matching input types ({names})")

    def execute(self, args, options=None):
        if len(args) != self.arity:
            raise ArrowInvalid(f"Function '{self.name}' accepts {self.arity} arguments but "
                               f"{len(args)} passed")
        options = options if options is not None else self.default_options
        if self.kind == Function.META:
            return self._meta_impl(args, options)
        if self.kind == Function.HASH_AGGREGATE:  # function.cc:325-326
            raise ArrowNotImplementedError("Direct execution of HASH_AGGREGATE functions")
        types = [a.type for a in args]
        return self.dispatch_exact(types).exec(args, options)


class FunctionRegistry:
    """registry.h:46: name -> Function, aliases, no overwrite unless asked."""

    def __init__(self):
        self._fns: dict[str, Function] = {}
        self._lock = threading.Lock()

    def add_function(self, fn: Function, allow_overwrite: bool = False) -> None:
        with self._lock:
            if fn.name in self._fns and not allow_overwrite:
                raise KeyError(f"Already have a function registered with name: {fn.name}")
            self._fns[fn.name] = fn

    def add_alias(self, target: str, source: str) -> None:
        with self._lock:
            if source not in self._fns:
                raise KeyError(f"No function registered with name: {source}")
            self._fns[target] = self._fns[source]

    def get_function(self, name: str) -> Function:
        try:
            return self._fns[name]
        except KeyError:
            raise KeyError(f"No function registered with name: {name}") from None

    def get_function_names(self):
        return sorted(self._fns)

    @property
    def num_functions(self) -> int:
        return len(self._fns)


# --------------------------------------------------------------------------- hash_sum kernel
class GroupedSumInt64State:
    """KernelState of hash_sum(int64, uint32): the device twin of
    GroupedReducingAggregator<Int64Type, GroupedSumImpl> (hash_aggregate_numeric.cc:44-187).
    Dense per-group arrays in HBM: sums, counts, null_seen (the complement of `no_nulls_`)."""

    def __init__(self, options: ScalarAggregateOptions | None, device):
        self.options = options or ScalarAggregateOptions()
        self.device = torch.device(device)
        self.num_groups = 0
        self.sums = torch.zeros(0, dtype=torch.int64, device=self.device)
        self.counts = torch.zeros(0, dtype=torch.int64, device=self.device)
        self.null_seen = torch.zeros(0, dtype=torch.int32, device=self.device)


def _hash_sum_init(options, device=None):
    """HashAggregateInit<Impl> (hash_aggregate_internal.h:54-63)."""
    from .array import default_device

    return GroupedSumInt64State(options, device if device is not None else default_device())


def _grow(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.numel() >= n:
        return t
    out = torch.zeros(max(n, 2 * t.numel()), dtype=t.dtype, device=t.device)
    out[: t.numel()].copy_(t)  # device-to-device buffer growth (TypedBufferBuilder::Append)
    return out


def _hash_sum_resize(state: GroupedSumInt64State, new_num_groups: int) -> None:
    """Resize (:61-68): new groups start at sum 0, count 0, no nulls seen."""
    state.sums = _grow(state.sums, new_num_groups)
    state.counts = _grow(state.counts, new_num_groups)
    state.null_seen = _grow(state.null_seen, new_num_groups)
    state.num_groups = new_num_groups


def _hash_sum_consume(state: GroupedSumInt64State, batch) -> None:
    """Consume (:70-83): batch = [values (Array | Scalar), group_ids (uint32 Array)]."""
    values, gids = batch[0], batch[1]
    if gids.type != uint32:
        raise ArrowInvalid("hash_sum: group ids must be uint32")
    lib, stream = _lib_and_stream(state.device)
    n = gids.length
    if isinstance(values, Scalar):
        span = _lib.ArxSpan(None, None, 0, n, 0 if values.is_valid else n)
        is_scalar, sv = 1, int(values.value or 0)
    else:
        if values.type != int64:
            raise ArrowNotImplementedError("hash_sum on the gfx950 path takes int64 values")
        if values.length != n:
            raise ArrowInvalid("Array arguments must all be the same length")
        span, is_scalar, sv = values.span(), 0, 0
    # with scratch the library partitions by group id and aggregates in LDS (large batches); without it every row is
    # a device atomic
    ws_bytes = lib.arx_hash_sum_consume_workspace_bytes(n, state.num_groups)
    ws_ptr, ws_len = None, 0
    if ws_bytes:
        ws = _workspace(state.device, ws_bytes + 256, "groupby")
        ws_ptr = (ws.data_ptr() + 255) & ~255
        ws_len = ws.numel() - (ws_ptr - ws.data_ptr())
    check(lib.arx_hash_sum_i64_consume_ws(C.byref(span), is_scalar, sv, gids.values_ptr(), n, state.num_groups,
                                          state.sums.data_ptr(), state.counts.data_ptr(),
                                          state.null_seen.data_ptr(), ws_ptr, ws_len, stream))


def _hash_sum_merge(state: GroupedSumInt64State, other: GroupedSumInt64State, group_id_mapping) -> None:
    """Merge (:85-107): group_id_mapping[other_g] = group id in `state` (uint32 Array)."""
    lib, stream = _lib_and_stream(state.device)
    g = group_id_mapping.length
    if g != other.num_groups:
        raise ArrowInvalid("group_id_mapping length must equal the other state's group count")
    check(lib.arx_hash_sum_i64_merge(state.sums.data_ptr(), state.counts.data_ptr(),
                                     state.null_seen.data_ptr(), other.sums.data_ptr(),
                                     other.counts.data_ptr(), other.null_seen.data_ptr(),
                                     group_id_mapping.values_ptr(), g, stream))


def _hash_sum_finalize(state: GroupedSumInt64State) -> Array:
    """Finalize (:130-152): int64 Array of num_groups sums; the validity bitmap is only
    materialised when a group is null (Finish :109-128) or skip_nulls is false."""
    lib, stream = _lib_and_stream(state.device)
    g = state.num_groups
    bits = alloc(bitmap_nbytes(g), state.device, zero=True)
    counter = torch.zeros(8, dtype=torch.uint8, device=state.device)
    check(lib.arx_hash_sum_i64_finalize(state.counts.data_ptr(), state.null_seen.data_ptr(), g,
                                        int(state.options.skip_nulls), state.options.min_count,
                                        bits.data_ptr(), counter.data_ptr(), stream))
    data = state.sums[:g].contiguous().view(torch.uint8)
    nulls = g - int(counter.cpu().view(torch.int64)[0])
    if state.options.skip_nulls and nulls == 0:
        return Array(int64, g, [None, data], 0, 0)
    return Array(int64, g, [bits, data], nulls if state.options.skip_nulls else kUnknownNullCount, 0)


class HashAggregateKernel:
    """compute/kernel.h:739-769: {signature, init, resize, consume, merge, finalize}."""

    def __init__(self, in_types, init, resize, consume, merge, finalize, out_type=None):
        self.in_types = tuple(in_types)
        self.init, self.resize, self.consume = init, resize, consume
        self.merge, self.finalize = merge, finalize
        self.out_type = out_type
        self.exec = None

    matches = Kernel.matches


class ExecBatch:
    """exec.h:174-261: a list of equal-length Arrays / Scalars plus the length."""

    def __init__(self, values, length: int | None = None):
        self.values = list(values)
        if length is None:
            lens = [v.length for v in self.values if isinstance(v, Array)]
            length = lens[0] if lens else 1
        self.length = length

    def __getitem__(self, i):
        return self.values[i]

    @property
    def num_values(self):
        return len(self.values)


class RecordBatch:
    """Just enough of arrow::RecordBatch for the Filter/Take meta-function shape dispatch."""

    def __init__(self, columns: dict):
        self.columns = dict(columns)
        lens = {c.length for c in self.columns.values()}
        if len(lens) > 1:
            raise ArrowInvalid("columns of a RecordBatch must have equal length")
        self.num_rows = lens.pop() if lens else 0


_FIXED_WIDTH = lambda t: t.bit_width >= 8  # noqa: E731
_INTEGER = lambda t: t.name in INDEX_TYPE_ID  # noqa: E731
_BASE_BINARY = lambda t: t.name in ("binary", "string")  # noqa: E731


def _filter_meta(args, options):
    """FilterMetaFunction::ExecuteImpl (vector_selection_filter_internal.cc:1043-1072)."""
    values, mask = args
    options = options or FilterOptions()
    if isinstance(values, RecordBatch):
        # FilterRecordBatch :925-960 — mask -> indices once, then Take per column
        if mask.length != values.num_rows:
            raise ArrowInvalid("Filter inputs must all be the same length")
        indices = get_take_indices(mask, options.null_selection_behavior)
        take_opts = TakeOptions(boundscheck=False)
        return RecordBatch({k: call_function("take", [c, indices], take_opts)
                            for k, c in values.columns.items()})
    if isinstance(mask, RunEndEncoded):
        mask = ree_mask_to_boolean(mask)
    return call_function("array_filter", [values, mask], options)


def ree_mask_to_boolean(mask: RunEndEncoded) -> Array:
    """A run_end_encoded<boolean> filter (the ree_filter input type of PopulateFilterKernels,
    vector_selection_filter_internal.cc:1090) expanded on the device into the plain boolean layout: every row carries
    its run's (valid, selected) pair, which is all VisitPlainxREEFilterOutputSegments reads."""
    if mask.values.type != bool_:
        raise ArrowNotImplementedError("filter: run-end-encoded masks must have boolean values")
    dev = mask.device
    lib, stream = _lib_and_stream(dev)
    n = mask.length
    bits = alloc(bitmap_nbytes(n), dev)
    has_nulls = mask.values.may_have_nulls()
    valid = alloc(bitmap_nbytes(n), dev) if has_nulls else None
    vs = mask.values.span()
    nruns = mask.run_ends.length
    re_ptr = mask.run_ends.values_ptr()
    check(lib.arx_ree_bool_expand(re_ptr, mask.run_ends.type.byte_width, nruns, C.byref(vs), mask.offset, n,
                                  bits.data_ptr(), None if valid is None else valid.data_ptr(), stream))
    return Array(bool_, n, [valid, bits], kUnknownNullCount if has_nulls else 0, 0)


def _take_meta(args, options):
    """TakeMetaFunction::ExecuteImpl (vector_selection_take_internal.cc:660-701), AAA and RAR."""
    values, indices = args
    options = options or TakeOptions()
    if isinstance(values, RecordBatch):
        # RAR (TakeRAR :619-633 runs TakeAAA per column): the fixed-width columns go through ONE launch that reads
        # the indices once (arx_take_columns, groups of 16); boolean / var-width columns keep their own kernels
        fused = {k: c for k, c in values.columns.items() if isinstance(c, Array) and c.type != bool_ and _FIXED_WIDTH(c.type)
                 and c.type.byte_width in (1, 2, 4, 8, 16, 32)}
        out = dict.fromkeys(values.columns)
        if len(fused) >= 2 and isinstance(indices, Array) and indices.type.name in INDEX_TYPE_ID:
            names = list(fused)
            for b in range(0, len(names), 16):
                part = names[b:b + 16]
                for k, col in zip(part, take_columns([fused[k] for k in part], indices, options.boundscheck)):
                    out[k] = col
        for k, c in values.columns.items():
            if out[k] is None:
                out[k] = call_function("array_take", [c, indices], options)
        return RecordBatch(out)
    return call_function("array_take", [values, indices], options)


def take_columns(columns, indices: Array, boundscheck: bool = True):
    """`take` of up to 16 fixed-width columns of equal length by one index array in one launch (arx_take_columns);
    per column the result of array_take.  One bounds check for all columns."""
    columns = list(columns)
    n = columns[0].length
    if any(c.length != n for c in columns):
        raise ArrowInvalid("columns of a RecordBatch must have equal length")
    dev = columns[0].device
    lib, stream = _lib_and_stream(dev)
    tid = INDEX_TYPE_ID[indices.type.name]
    ispan = indices.span()
    if boundscheck:
        ws = _workspace(dev, lib.arx_take_workspace_bytes(), "take")
        check(lib.arx_check_index_bounds(C.byref(ispan), tid, n, ws.data_ptr(), ws.numel(), stream))
    m, k = indices.length, len(columns)
    spans = (_lib.ArxSpan * k)(*[c.span() for c in columns])
    widths = (C.c_int32 * k)(*[c.type.byte_width for c in columns])
    out_data = [alloc(m * c.type.byte_width, dev) for c in columns]
    need = [c.may_have_nulls() or indices.may_have_nulls() for c in columns]
    out_valid = [alloc(bitmap_nbytes(m), dev) if nd else None for nd in need]
    counters = torch.zeros(k, dtype=torch.int64, device=dev)
    data_ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in out_data])
    valid_ptrs = (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in out_valid])
    with tracing.span("arx_take_columns"):
        check(lib.arx_take_columns(spans, widths, k, C.byref(ispan), tid, data_ptrs, valid_ptrs, counters.data_ptr(),
                                   stream))
    outs = []
    for j, c in enumerate(columns):
        a = Array(c.type, m, [out_valid[j], out_data[j]], 0, 0)
        if need[j]:
            a.set_lazy_null_count(_LazyColumnCount(m, counters, j))
        outs.append(a)
    return outs


class _LazyColumnCount:
    """null_count of column j from the per-column valid counters of arx_take_columns (read on first use)."""

    def __init__(self, length, counters, j):
        self.length, self.counters, self.j = length, counters, j

    def __call__(self):
        return self.length - int(self.counters[self.j].item())


def _cast_meta(args, options):
    """CastMetaFunction::ExecuteImpl (cast.cc:95-126): dispatch on the target type."""
    (arr,) = args
    if options is None or options.to_type is None:
        raise ArrowInvalid("Cast requires that options.to_type is set")
    if arr.type == options.to_type:
        return arr
    return _cast_table_lookup(options.to_type).execute([arr], options)


_cast_table: dict[str, Function] = {}


def _cast_table_lookup(to_type: DataType) -> Function:
    """GetCastFunction (cast.cc:207-214): casts live in a private table, not the registry."""
    try:
        return _cast_table[to_type.name]
    except KeyError:
        raise ArrowNotImplementedError(f"Unsupported cast to {to_type.name} (no available cast "
                                       "function for target type)") from None


def _sort_indices_meta(args, options):
    """SortIndicesMetaFunction::ExecuteImpl (vector_sort.cc:856-924) for Array input."""
    (arr,) = args
    options = options or SortOptions()
    order = options.sort_keys[0][1] if options.sort_keys else "ascending"
    return call_function("array_sort_indices", [arr], ArraySortOptions(order, options.null_placement))


def _build_registry() -> FunctionRegistry:
    reg = FunctionRegistry()

    f = Function("array_filter", Function.VECTOR, 2, FilterOptions())
    f.add_kernel(Kernel((_FIXED_WIDTH, bool_), _exec_array_filter))
    f.add_kernel(Kernel((bool_, bool_), _exec_boolean_filter))
    f.add_kernel(Kernel((_BASE_BINARY, bool_), _exec_binary_filter))
    reg.add_function(f)
    reg.add_function(Function("filter", Function.META, 2, FilterOptions(), _filter_meta))

    f = Function("array_take", Function.VECTOR, 2, TakeOptions())
    f.add_kernel(Kernel((_FIXED_WIDTH, _INTEGER), _exec_array_take))
    f.add_kernel(Kernel((bool_, _INTEGER), _exec_array_take))
    f.add_kernel(Kernel((_BASE_BINARY, _INTEGER), _exec_binary_take))
    reg.add_function(f)
    reg.add_function(Function("take", Function.META, 2, TakeOptions(), _take_meta))

    # one cast function per numeric target (GetCastFunction, cast.cc:207-214; kernels of
    # scalar_cast_numeric.cc:797-858): the generic pair kernel first, the tuned ones after it (last match wins)
    for to in (int8, uint8, int16, uint16, int32, uint32, int64, uint64, float32, float64):
        c = Function("cast_" + to.name, Function.SCALAR, 1)
        c.add_kernel(Kernel((_NUMERIC,), _make_exec_cast_numeric(to), to))
        _cast_table[to.name] = c
    _cast_table["float"].add_kernel(Kernel((float64,), _exec_cast_f64_f32, float32))
    _cast_table["int32"].add_kernel(Kernel((int64,), _exec_cast_i64_i32, int32))
    _cast_table["int64"].add_kernel(Kernel((int32,), _exec_cast_i32_i64, int64))
    _cast_table["double"].add_kernel(Kernel((int64,), _exec_cast_i64_f64, float64))
    reg.add_function(Function("cast", Function.META, 1, None, _cast_meta))

    from .array import type_from_name
    numeric_types = [type_from_name(nm) for nm in _NUM_TYPE_ID]
    f = Function("greater", Function.SCALAR, 2)
    for t in numeric_types:      # every numeric type first; the tuned 64-bit kernels are added after them and win
        f.add_kernel(Kernel((t, t), _exec_compare_numeric("greater"), bool_))
    f.add_kernel(Kernel((float64, float64), _exec_greater, bool_))
    f.add_kernel(Kernel((int64, int64), _exec_greater, bool_))
    reg.add_function(f)
    # temporal operands (timestamp / duration / time / date): the comparisons of their physical integers; both sides of
    # the same type and unit — the reference unifies units by implicit casts before dispatch (DispatchBest), and compares
    # a zoned timestamp only with a zoned one (scalar_compare.cc:299-313)
    from .array import is_temporal

    def _temporal_pair_kernel(name):
        run = _exec_compare_numeric(name)

        def checked(args, options):
            types = [a.type for a in args if isinstance(a, (Array, Scalar))]
            if len({t.name.split(", tz=")[0].rstrip("]") for t in types}) > 1:
                raise ArrowNotImplementedError(f"{name}: temporal operands of different types / units ({', '.join(t.name for t in types)})")
            zoned = {", tz=" in t.name for t in types if t.name.startswith("timestamp")}
            if len(zoned) > 1:
                raise ArrowInvalid("Cannot compare timestamp with timezone to timestamp without timezone, got: "
                                   + " and ".join(t.name for t in types))
            return run(args, options)
        return Kernel((is_temporal, is_temporal), checked, bool_)


    for name in ("equal", "not_equal", "greater_equal", "less", "less_equal"):
        f = Function(name, Function.SCALAR, 2)
        for t in numeric_types:
            f.add_kernel(Kernel((t, t), _exec_compare_numeric(name), bool_))
        f.add_kernel(Kernel((float64, float64), _exec_compare(name), bool_))
        f.add_kernel(Kernel((int64, int64), _exec_compare(name), bool_))
        f.add_kernel(_temporal_pair_kernel(name))
        reg.add_function(f)
    reg.get_function("greater").add_kernel(_temporal_pair_kernel("greater"))

    f = Function("add", Function.SCALAR, 2)
    for t in numeric_types:
        f.add_kernel(Kernel((t, t), _exec_arith_numeric("add", False), t))
    f.add_kernel(Kernel((int64, int64), _exec_add, int64))
    f.add_kernel(Kernel((float64, float64), _exec_add, float64))
    reg.add_function(f)
    for name, op, checked in (("subtract", "subtract", False), ("multiply", "multiply", False),
                              ("add_checked", "add", True), ("subtract_checked", "subtract", True),
                              ("multiply_checked", "multiply", True)):
        f = Function(name, Function.SCALAR, 2)
        for t in numeric_types:
            f.add_kernel(Kernel((t, t), _exec_arith_numeric(op, checked), t))
        f.add_kernel(Kernel((int64, int64), _exec_arith(op, checked), int64))
        f.add_kernel(Kernel((float64, float64), _exec_arith(op, checked), float64))
        reg.add_function(f)

    for name, checked in (("divide", False), ("divide_checked", True)):
        f = Function(name, Function.SCALAR, 2)
        for t in numeric_types:
            f.add_kernel(Kernel((t, t), _exec_divide_numeric(checked), t))
        f.add_kernel(Kernel((int64, int64), _exec_divide(checked), int64))
        f.add_kernel(Kernel((float64, float64), _exec_divide(checked), float64))
        reg.add_function(f)

    for name, code in (("and_kleene", 0), ("or_kleene", 1)):
        f = Function(name, Function.SCALAR, 2)
        f.add_kernel(Kernel((bool_, bool_), _exec_kleene(code), bool_))
        reg.add_function(f)
    f = Function("invert", Function.SCALAR, 1)
    f.add_kernel(Kernel((bool_,), _exec_invert, bool_))
    reg.add_function(f)

    f = Function("array_sort_indices", Function.VECTOR, 1, ArraySortOptions())
    for key_type in (uint64, int64, uint32, int32, float64, float32):
        f.add_kernel(Kernel((key_type,), _exec_array_sort_indices, uint64))
    f.add_kernel(Kernel((bool_,), _exec_array_sort_indices_bool, uint64))
    reg.add_function(f)
    reg.add_function(Function("sort_indices", Function.META, 1, SortOptions(), _sort_indices_meta))

    f = Function("hash_sum", Function.HASH_AGGREGATE, 2, ScalarAggregateOptions())
    f.add_kernel(HashAggregateKernel((int64, uint32), _hash_sum_init, _hash_sum_resize,
                                     _hash_sum_consume, _hash_sum_merge, _hash_sum_finalize, int64))
    reg.add_function(f)
    return reg


_registry = None


def get_function_registry() -> FunctionRegistry:
    """GetFunctionRegistry (registry.cc:301-304): the process-wide singleton."""
    global _registry
    if _registry is None:
        _registry = _build_registry()
    return _registry


def call_function(name: str, args, options=None, registry: FunctionRegistry | None = None):
    """CallFunction (exec.cc:1362-1370): registry lookup + Function::Execute."""
    reg = registry or get_function_registry()
    return reg.get_function(name).execute(list(args), options)


# --------------------------------------------------------------------------- typed wrappers
def filter(values, mask, null_selection_behavior: str = "drop"):  # noqa: A001
    """compute::Filter (api_vector.cc:412-416)."""
    return call_function("filter", [values, mask], FilterOptions(null_selection_behavior))


def take(values, indices, boundscheck: bool = True):
    """compute::Take (api_vector.cc:419-423)."""
    return call_function("take", [values, indices], TakeOptions(boundscheck))


def cast(arr, to_type: DataType, safe: bool = True):
    """compute::Cast (cast.cc:239-248); safe=False = CastOptions::Unsafe."""
    return call_function("cast", [arr], CastOptions(to_type, allow_float_truncate=not safe, allow_int_overflow=not safe))


def _wrap_scalar(x, like: Array):
    if isinstance(x, (Array, Scalar)):
        return x
    return Scalar(x, like.type, x is not None)


def greater(left, right):
    like = left if isinstance(left, Array) else right
    left, right = _wrap_scalar(left, like), _wrap_scalar(right, like)
    return call_function("greater", [left, right])


def _compare_wrapper(name):
    def fn(left, right):
        like = left if isinstance(left, Array) else right
        return call_function(name, [_wrap_scalar(left, like), _wrap_scalar(right, like)])
    fn.__name__ = name
    fn.__doc__ = f"compute::CallFunction(\"{name}\") (scalar_compare.cc:391-445)."
    return fn


equal, not_equal, greater_equal, less, less_equal = (_compare_wrapper(n) for n in
                                                     ("equal", "not_equal", "greater_equal", "less", "less_equal"))
subtract, multiply, add_checked, subtract_checked, multiply_checked = (
    _compare_wrapper(n) for n in ("subtract", "multiply", "add_checked", "subtract_checked", "multiply_checked"))
divide, divide_checked = (_compare_wrapper(n) for n in ("divide", "divide_checked"))


def add(left, right):
    like = left if isinstance(left, Array) else right
    left, right = _wrap_scalar(left, like), _wrap_scalar(right, like)
    return call_function("add", [left, right])


def and_kleene(left, right):
    return call_function("and_kleene", [left, right])


def or_kleene(left, right):
    return call_function("or_kleene", [left, right])


def invert(arr):
    return call_function("invert", [arr])


def sort_indices(arr, order: str = "ascending", null_placement: str = "at_end"):
    """compute::SortIndices (api_vector.cc:334-347)."""
    return call_function("sort_indices", [arr], SortOptions([("", order)], null_placement))


def concat_arrays(arrays) -> Array:
    """arrow::Concatenate (array/concatenate.cc: ConcatenateBitmaps, ConcatenateBuffers, PutOffsets) for
    fixed-width, boolean and utf8/binary arrays: one output array, chunks glued at arbitrary bit positions.
    What an operator that must see its whole input does first (OrderByNode, acero/order_by_node.cc:100-108)."""
    arrays = list(arrays)
    if not arrays:
        raise ArrowInvalid("Must pass at least one array")
    t = arrays[0].type
    if any(a.type != t for a in arrays):
        raise ArrowInvalid("arrays to be concatenated must be identically typed")
    dev = arrays[0].device
    lib, stream = _lib_and_stream(dev)
    n = builtins.sum(a.length for a in arrays)
    have_validity = any(a.validity is not None and a.null_count != 0 for a in arrays)
    out_valid = alloc(bitmap_nbytes(n), dev, zero=True) if have_validity else None
    pos = 0
    if have_validity:
        for a in arrays:
            check(lib.arx_bitmap_copy_at((a.validity.data_ptr() if a.validity is not None and a.null_count != 0 else None), a.offset, a.length,
                                         out_valid.data_ptr(), pos, stream))
            pos += a.length
    if is_base_binary(t):
        # (two offsets per chunk come back to the host: where its bytes start and end)
        ends = [a.buffers[1].view(torch.int32)[[a.offset, a.offset + a.length]].cpu().tolist() if a.length else [0, 0]
                for a in arrays]
        total = builtins.sum(e - b for b, e in ends)
        if total > 2**31 - 1:
            raise ArrowInvalid("offset overflow while concatenating arrays")
        out_off = alloc((n + 1) * 4, dev, zero=True)
        out_data = alloc(total, dev)
        pos, base = 0, 0
        for a, (b, e) in zip(arrays, ends):
            if a.length == 0:
                continue
            check(lib.arx_binary_rebase_offsets(a.buffers[1].data_ptr() + a.offset * 4, a.length, base,
                                                out_off.data_ptr() + pos * 4, stream))
            if e > b:
                out_data[base:base + e - b].copy_(a.buffers[2][b:e])
            pos += a.length
            base += e - b
        out = Array(t, n, [out_valid, out_off, out_data], kUnknownNullCount if have_validity else 0, 0)
    elif t == bool_:
        out_bits = alloc(bitmap_nbytes(n), dev, zero=True)
        pos = 0
        for a in arrays:
            check(lib.arx_bitmap_copy_at(a.data.data_ptr(), a.offset, a.length, out_bits.data_ptr(), pos, stream))
            pos += a.length
        out = Array(t, n, [out_valid, out_bits], kUnknownNullCount if have_validity else 0, 0)
    else:
        w = t.byte_width
        out_data = alloc(n * w, dev)
        pos = 0
        for a in arrays:
            out_data[pos * w:(pos + a.length) * w].copy_(a.data[a.offset * w:(a.offset + a.length) * w])
            pos += a.length
        out = Array(t, n, [out_valid, out_data], kUnknownNullCount if have_validity else 0, 0)
    if have_validity:
        known = [a.null_count for a in arrays]
        if all(k >= 0 for k in known):
            out._null_count = builtins.sum(known)
    return out


def sort_indices_by_keys(keys, orders=None, null_placement="at_end") -> Array:
    """SortIndices(Table, SortOptions) for several sort keys (TableSorter / MultipleKeyRecordBatchSorter,
    kernels/vector_sort.cc:850-954): lexicographic, stable, every key with its own order and its own null
    placement (SortKey::null_placement, compute/ordering.h:50-61; one string = the SortOptions-wide override,
    api_vector.h:132-143).  Least-significant key first, one stable device sort per key: perm = sort(k_last);
    then perm = take(perm, sort(take(k, perm))) for each earlier key."""
    keys = list(keys)
    orders = list(orders) if orders is not None else ["ascending"] * len(keys)
    places = [null_placement] * len(keys) if isinstance(null_placement, str) else list(null_placement)
    if not keys or len(keys) != len(orders) or len(keys) != len(places):
        raise ArrowInvalid("Must specify one or more sort keys")
    perm = None
    for key, order, place in zip(reversed(keys), reversed(orders), reversed(places)):
        opts = ArraySortOptions(order, place)
        if perm is None:
            perm = call_function("array_sort_indices", [key], opts)
        else:
            step = call_function("array_sort_indices", [take(key, perm, boundscheck=False)], opts)
            perm = take(perm, step, boundscheck=False)
    return perm


# RankOptions::Tiebreaker (api_vector.h:201-212) = ARX_RANK_* of include/arrow_amd.h; "quantile": rank_quantile, "normal":
# rank_normal
_RANK_TIEBREAKER = {"min": 0, "max": 1, "first": 2, "dense": 3, "quantile": 4, "normal": 5}


def _rank(arr: Array, order: str, null_placement: str, tiebreaker: str) -> Array:
    """RankMetaFunctionBase::Rank (kernels/vector_rank.cc:386-420): sort the indices with the array sorter, then one
    walk of the sorted order (arx_rank)."""
    if arr.type.name not in _SORT_KEY_TYPE:
        raise ArrowNotImplementedError(f"rank: {arr.type.name} values")
    if tiebreaker not in _RANK_TIEBREAKER:
        raise ArrowInvalid(f"rank: tiebreaker {tiebreaker!r}")
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    n = arr.length
    sorted_rows = call_function("array_sort_indices", [arr], ArraySortOptions(order, null_placement))
    out = alloc(n * 8, dev)
    ws_bytes = lib.arx_rank_workspace_bytes(n)
    ws = _workspace(dev, ws_bytes + 256, "rank")
    base = ws.data_ptr()
    aligned = (base + 255) & ~255
    span = arr.span()
    check(lib.arx_rank(C.byref(span), _SORT_KEY_TYPE[arr.type.name], sorted_rows.data.data_ptr(), _RANK_TIEBREAKER[tiebreaker],
                       aligned, ws.numel() - (aligned - base), out.data_ptr(), stream))
    return Array(float64 if tiebreaker in ("quantile", "normal") else uint64, n, [None, out], 0, 0)


def rank(arr: Array, order: str = "ascending", null_placement: str = "at_end", tiebreaker: str = "first") -> Array:
    """compute "rank" (RankMetaFunction, kernels/vector_rank.cc:422-441; RankOptions api_vector.h:195-240): 1-based uint64
    ranks, never null; nulls and NaNs are ranked where the sort puts them, all nulls tie, all NaNs tie."""
    return _rank(arr, order, null_placement, tiebreaker)


def rank_quantile(arr: Array, order: str = "ascending", null_placement: str = "at_end") -> Array:
    """compute "rank_quantile" (RankQuantileMetaFunction, kernels/vector_rank.cc:443-462): (rows below the row's run of
    ties + half the run) / length, float64."""
    return _rank(arr, order, null_placement, "quantile")


def rank_normal(arr: Array, order: str = "ascending", null_placement: str = "at_end") -> Array:
    """compute "rank_normal" (RankNormalMetaFunction, kernels/vector_rank.cc:424-441): the normal percent-point function
    (NormalPPF, util/math_internal.cc:26-137 — Wichura's AS 241) of the quantile rank, float64."""
    return _rank(arr, order, null_placement, "normal")


def partition_nth_indices(arr: Array, pivot: int, null_placement: str = "at_end") -> Array:
    """compute "partition_nth_indices" (PartitionNthToIndices, kernels/vector_array_sort.cc:56-95): uint64 indices such
    that the `pivot` first ones point at values no greater than the others; nulls and NaNs at the chosen end.  The
    reference runs std::nth_element (any valid partition is a right answer); a full sort is one such partition."""
    if pivot > arr.length:
        raise ArrowIndexError("NthToIndices index out of bound")
    return call_function("array_sort_indices", [arr], ArraySortOptions("ascending", null_placement))


# select_k by a THRESHOLD instead of a sort (round 6): rows below which row counts the fast path is not tried, the largest
# k / n it is tried for, and the largest share of the rows the candidates may be before the sort is the better plan
SELECT_K_MIN_ROWS = 1 << 22
SELECT_K_MAX_SHARE = 16
_SELECT_COUNTERS = {"threshold": 0, "sorted": 0}


def _select_k_by_threshold(arr: Array, k: int, order: str, null_placement: str):
    """The first k rows of the sort order WITHOUT sorting the column (uint64 / int64 keys, k much smaller than the
    column, every selected row a non-null): one pass for a 4096-bin histogram of the keys over their (sampled, widened)
    range — the pieces the sharded sort places its splitters with —, the bin in which the k-th smallest key falls, ONE
    comparison of the column against that bin's upper edge (`less_equal` / `greater_equal`: a bitmap), GetTakeIndices
    of the bitmap (the candidates' rows, ascending), and the sort of the candidates' keys alone.  The candidates are
    exactly the rows whose bin is at most that bin (the bin is monotone in the key, keys outside the sampled range
    included: they saturate into the end bins), they come in row order and the sort is stable, so the result is INDEX
    for index the head of the full stable sort.  ~3 passes over 8-byte keys instead of the sort's 51 B/row.
    None: not this case (the caller sorts)."""
    from .array import int64 as i64, uint64 as u64

    n = arr.length
    if arr.type not in (i64, u64) or n < SELECT_K_MIN_ROWS or n > 0xFFFFFFFF or k <= 0 or k * SELECT_K_MAX_SHARE > n:
        return None
    nulls = arr.null_count if (arr.buffers[0] is not None) else 0
    if nulls and (null_placement == "at_start" or k > n - nulls):
        return None
    dev = arr.device
    lib, stream = _lib_and_stream(dev)
    span = arr.span()
    is_signed = int(arr.type == i64)
    order_code = _lib.SORT_DESCENDING if order == "descending" else _lib.SORT_ASCENDING
    key_range = torch.zeros(2, dtype=torch.int64, device=dev)
    check(lib.arx_sort_key_range_sampled(C.byref(span), is_signed, order_code, 4, key_range.data_ptr(), stream))
    mask64 = (1 << 64) - 1
    inv_min, key_max = [int(x) & mask64 for x in key_range.tolist()]
    key_min = ~inv_min & mask64
    if key_max <= key_min:
        return None
    margin = ((key_max - key_min) >> 6) + 1
    key_min, key_max = max(0, key_min - margin), min(mask64, key_max + margin)
    shift = 64 - (key_max - key_min).bit_length()
    window = _lib.ArxSortKeyWindow(key_min, shift, 0)
    bits = 12
    hist = torch.zeros(1 << bits, dtype=torch.int64, device=dev)
    check(lib.arx_sort_key_histogram_window(C.byref(span), is_signed, order_code, bits, C.byref(window), hist.data_ptr(), stream))
    cum = torch.cumsum(hist.cpu(), 0)
    b = int(torch.searchsorted(cum, torch.tensor(k, dtype=cum.dtype)).item())
    if b >= (1 << bits) - 1 or int(cum[b]) * 4 > n:      # the last bin (open above), or a bin that holds a fifth of the column
        return None
    edge = key_min + ((((b + 1) << (64 - bits)) - 1) >> shift)      # the largest transformed key whose bin is at most b
    if edge >= mask64:
        return None
    # transformed key -> the column's value: descending sorts ~key, signed keys have their sign bit flipped
    raw = edge ^ mask64 if order == "descending" else edge
    if is_signed:
        raw ^= 1 << 63
        raw = raw - (1 << 64) if raw >= (1 << 63) else raw
    bound = Scalar(raw, arr.type, True)
    mask = call_function("greater_equal" if order == "descending" else "less_equal", [arr, bound])
    rows = get_take_indices(mask)               # nulls dropped; ascending row numbers
    if rows.length != int(cum[b]):
        raise ArrowInvalid(f"select_k_unstable: {rows.length} candidates, the histogram promised {int(cum[b])}")
    cand = take(arr, rows, boundscheck=False)
    perm = call_function("array_sort_indices", [cand], ArraySortOptions(order, "at_end"))
    head = Array(uint64, k, [None, perm.data[: k * 8]], 0, 0)
    picked = take(rows, head, boundscheck=False)
    _SELECT_COUNTERS["threshold"] += 1
    return picked if picked.type == uint64 else cast(picked, uint64)


def select_k_unstable(arr: Array, k: int, order: str = "ascending", null_placement: str = "at_end") -> Array:
    """compute "select_k_unstable" on an array (ArraySelector, kernels/vector_select_k.cc:156-232): the indices of the
    first k rows of the sorted order — non-nulls, NaNs, nulls taken in that sequence (or the reverse, at_start:
    CalculateOutputRangesByNullLikeness, :82-100).  "Unstable": the order of equal values is not promised; this one
    keeps them in row order."""
    if k < 0:
        raise ArrowInvalid("select_k_unstable requires a nonnegative `k`, got " + str(k))
    fast = _select_k_by_threshold(arr, k, order, null_placement)
    if fast is not None:
        return fast
    _SELECT_COUNTERS["sorted"] += 1
    perm = call_function("array_sort_indices", [arr], ArraySortOptions(order, null_placement))
    k = min(k, arr.length)
    return Array(uint64, k, [None, perm.data[: k * 8]], 0, 0)


def order_by(columns, sort_keys, null_placement="at_end"):
    """OrderByNode::DoFinish (acero/order_by_node.cc:100-108): concatenate the accumulated batches,
    SortIndices on the sort keys, Take on every column.  `columns` = list of columns, each a list of chunks;
    `sort_keys` = [(column index, "ascending" | "descending")]."""
    whole = [concat_arrays(chunks) if len(chunks) != 1 or chunks[0].offset != 0 else chunks[0] for chunks in columns]
    perm = sort_indices_by_keys([whole[i] for i, _ in sort_keys], [o for _, o in sort_keys], null_placement)
    return [take(col, perm, boundscheck=False) for col in whole]


# --------------------------------------------------------------------------- group-by
# --------------------------------------------------------------------------- scalar aggregates (int64)
class Int64Aggregator:
    """SumImpl / CountImpl / MinMaxImpl over int64 batches (aggregate_basic.inc.cc:49-110,776-860):
    consume() any number of device arrays, then sum() / count() / min_max() apply the options the
    way Finalize does.  (mean is not offered: the reference accumulates it in doubles, in row order.)"""

    def __init__(self, device=None, options: ScalarAggregateOptions | None = None):
        from .array import default_device

        self.device = torch.device(device) if device is not None else default_device()
        self.options = options or ScalarAggregateOptions()
        self.acc = torch.zeros(4, dtype=torch.int64, device=self.device)
        self.rows = 0
        self.nulls_observed = False
        lib, stream = _lib_and_stream(self.device)
        check(lib.arx_reduce_i64_init(self.acc.data_ptr(), stream))

    def consume(self, arr: Array) -> None:
        if arr.type != int64:
            raise ArrowNotImplementedError("Int64Aggregator: int64 values only")
        lib, stream = _lib_and_stream(self.device)
        sp = arr.span()
        check(lib.arx_reduce_i64_consume(C.byref(sp), self.acc.data_ptr(), stream))
        self.rows += arr.length

    def _state(self):
        s, c, mn, mx = (int(x) for x in self.acc.cpu().tolist())
        return s, c, mn, mx

    def sum(self):
        s, c, _, _ = self._state()
        nulls = c < self.rows
        if (not self.options.skip_nulls and nulls) or c < self.options.min_count:
            return None
        return s

    def count(self, mode: str = "only_valid"):
        _, c, _, _ = self._state()
        return {"only_valid": c, "only_null": self.rows - c, "all": self.rows}[mode]

    def min_max(self):
        _, c, mn, mx = self._state()
        nulls = c < self.rows
        if (not self.options.skip_nulls and nulls) or c < max(1, self.options.min_count):
            return None
        return mn, mx


def sum(arr: Array, skip_nulls: bool = True, min_count: int = 1):  # noqa: A001
    """compute::Sum for int64 (wrap-around), None = null."""
    agg = Int64Aggregator(arr.device, ScalarAggregateOptions(skip_nulls, min_count))
    agg.consume(arr)
    return agg.sum()


def count(arr: Array, mode: str = "only_valid") -> int:
    agg = Int64Aggregator(arr.device)
    agg.consume(arr)
    return agg.count(mode)


def min_max(arr: Array, skip_nulls: bool = True, min_count: int = 1):
    agg = Int64Aggregator(arr.device, ScalarAggregateOptions(skip_nulls, min_count))
    agg.consume(arr)
    return agg.min_max()


def _next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p


class GroupBySum:
    """The fused device operator standing where Acero's GroupByNode drives
    Grouper::Consume + hash_sum {resize, consume, merge, finalize}
    (acero/groupby_aggregate_node.cc:210-337, compute/kernel.h:720-769).

    capacity: slots of the open-addressing table (power of two, > expected distinct keys).
    """

    def __init__(self, capacity: int, device=None, options: ScalarAggregateOptions | None = None):
        from .array import default_device

        self.device = torch.device(device) if device is not None else default_device()
        self.capacity = _next_pow2(max(2, int(capacity)))
        self.options = options or ScalarAggregateOptions()
        lib, stream = _lib_and_stream(self.device)
        self.state = alloc(lib.arx_groupby_state_bytes(self.capacity), self.device)
        check(lib.arx_groupby_init(self.state.data_ptr(), self.capacity, stream))

    # ---- key / value types beyond (int32, int64): what the reference registers (Grouper key types,
    # row/grouper.cc:559-611; hash_sum value types, hash_aggregate_numeric.cc:1188-1200) mapped onto the one device
    # table by width-changing casts that are bijections on the values present
    _KEY32_SIGNED = ("int8", "int16", "int32", "date32[day]", "time32[s]", "time32[ms]")
    _KEY32_UNSIGNED = ("uint8", "uint16", "uint32")

    def _normalise_key(self, keys: Array) -> Array:
        name = keys.type.name
        prev = getattr(self, "key_type", None)
        if prev is not None and prev != keys.type:
            raise ArrowInvalid(f"GroupBySum: key type changed from {prev.name} to {name}")
        self.key_type = keys.type
        if keys.type == int32:
            return keys
        if name in ("uint32", "date32[day]", "time32[s]", "time32[ms]"):
            return Array(int32, keys.length, keys.buffers, keys.null_count, keys.offset)      # same bits, same equality
        if name in ("int8", "int16", "uint8", "uint16"):
            return cast(keys, int32)                                                          # value-preserving widening
        if keys.type.bit_width == 64 and keys.type.name != "double":
            # 64-bit integer / temporal keys: the device table holds 32-bit keys, so they are accepted when every
            # valid key fits (a checked cast: the first one that does not names the error), declined otherwise
            try:
                if name == "uint64":
                    return cast(keys, int32)
                as_i64 = keys if keys.type == int64 else Array(int64, keys.length, keys.buffers, keys.null_count, keys.offset)
                return cast(as_i64, int32)
            except ArrowInvalid as e:
                raise ArrowNotImplementedError(f"GroupBySum: {name} keys beyond the int32 range are not implemented "
                                               f"(the device table holds 32-bit keys): {e}") from None
        raise ArrowNotImplementedError(f"GroupBySum: keys of type {name}")

    def _typed_keys(self, keys_i32: torch.Tensor) -> torch.Tensor:
        """The table's int32 keys in the CALLER's key type: the bit-reinterpreted 32-bit types get their own
        view back (uint32 >= 2^31 would read as negative), the widened narrow integers are narrowed again (every value
        fits: it came from that type), 64-bit keys that passed the checked cast are widened.  Temporal 32-bit types
        keep their physical int32 values (days / seconds / milliseconds), 64-bit temporal types int64."""
        kt = getattr(self, "key_type", None)
        if kt is None or kt == int32:
            return keys_i32
        name = kt.name
        if name == "uint32":
            return keys_i32.view(torch.uint32)
        if name in ("date32[day]", "time32[s]", "time32[ms]"):
            return keys_i32
        narrow = {"int8": torch.int8, "int16": torch.int16, "uint8": torch.uint8, "uint16": torch.uint16}
        if name in narrow:
            return keys_i32.to(narrow[name])
        if name == "uint64":
            return keys_i32.to(torch.int64).view(torch.uint64)      # (non-negative: it passed the checked cast)
        return keys_i32.to(torch.int64)

    def _normalise_value(self, values: Array) -> Array:
        name = values.type.name
        prev = getattr(self, "value_type", None)
        if prev is not None and prev != values.type:
            raise ArrowInvalid(f"GroupBySum: value type changed from {prev.name} to {name}")
        self.value_type = values.type
        if values.type == int64:
            return values
        if name == "uint64":      # the accumulator of unsigned inputs is UInt64 (FindAccumulatorType): same wrap-around bits
            return Array(int64, values.length, values.buffers, values.null_count, values.offset)
        if name in ("int8", "int16", "int32"):
            return cast(values, int64)
        if name in ("uint8", "uint16", "uint32"):
            w = cast(values, uint64)
            return Array(int64, w.length, w.buffers, w.null_count, w.offset)
        if name in ("float", "double"):
            raise ArrowNotImplementedError(
                f"GroupBySum: hash_sum({name}) accumulates doubles in row order in the reference "
                "(hash_aggregate_numeric.cc:70-83), which a parallel reduction cannot reproduce bit for bit")
        raise ArrowNotImplementedError(f"GroupBySum: values of type {name}")

    @property
    def sum_type(self) -> DataType:
        """Output type of hash_sum for the consumed value type (FindAccumulatorType, aggregate_internal.h:41-44)."""
        vt = getattr(self, "value_type", int64)
        return uint64 if vt.name.startswith("uint") else int64

    def consume(self, keys: Array, values: Array) -> None:
        keys, values = self._normalise_key(keys), self._normalise_value(values)
        lib, stream = _lib_and_stream(self.device)
        ks, vs = keys.span(), values.span()
        ws_bytes = lib.arx_groupby_consume_workspace_bytes(keys.length, self.capacity)
        ws_ptr, ws_len = None, 0
        if ws_bytes:
            ws = _workspace(self.device, ws_bytes + 256, "groupby")
            ws_ptr = (ws.data_ptr() + 255) & ~255
            ws_len = ws.numel() - (ws_ptr - ws.data_ptr())
        check(lib.arx_groupby_sum_i64_consume(self.state.data_ptr(), self.capacity, C.byref(ks),
                                              C.byref(vs), ws_ptr, ws_len, stream))

    # -- hash_min / hash_max on the same table (GroupedMinMaxImpl, kernels/hash_aggregate.cc:330-419)
    def consume_min_max(self, keys: Array, values: Array) -> None:
        """Folds the rows into per-group extrema; does not touch the sums, so the same rows may also
        go through consume()."""
        keys = self._normalise_key(keys)
        if values.type != int64:
            raise ArrowNotImplementedError("GroupBySum: hash_min / hash_max over int64 values only")
        lib, stream = _lib_and_stream(self.device)
        if getattr(self, "minmax", None) is None:
            self.minmax = alloc(lib.arx_groupby_minmax_bytes(self.capacity), self.device)
            check(lib.arx_groupby_minmax_init(self.minmax.data_ptr(), self.capacity, stream))
        ks, vs = keys.span(), values.span()
        check(lib.arx_groupby_minmax_i64_consume(self.state.data_ptr(), self.minmax.data_ptr(), self.capacity,
                                                 C.byref(ks), C.byref(vs), stream))

    def export_min_max(self):
        """Partial extrema: dict keys / key_is_valid / mins / maxs / no_nulls (device tensors)."""
        lib, stream = _lib_and_stream(self.device)
        g = self.num_groups()
        dev = self.device
        t = lambda dt: torch.empty(max(g, 1), dtype=dt, device=dev)  # noqa: E731
        cols = dict(keys=t(torch.int32), key_is_valid=t(torch.uint8), sums=t(torch.int64), counts=t(torch.int64),
                    no_nulls=t(torch.uint8), mins=t(torch.int64), maxs=t(torch.int64))
        check(lib.arx_groupby_export(self.state.data_ptr(), self.minmax.data_ptr(), cols["keys"].data_ptr(),
                                     cols["key_is_valid"].data_ptr(), cols["sums"].data_ptr(),
                                     cols["counts"].data_ptr(), cols["no_nulls"].data_ptr(),
                                     cols["mins"].data_ptr(), cols["maxs"].data_ptr(), stream))
        return {k: v[:g] for k, v in cols.items()}

    def merge_min_max(self, partial: dict) -> None:
        """Merge of GroupedMinMaxImpl (hash_aggregate.cc:371-399) from export_min_max() of another state."""
        lib, stream = _lib_and_stream(self.device)
        if getattr(self, "minmax", None) is None:
            self.minmax = alloc(lib.arx_groupby_minmax_bytes(self.capacity), self.device)
            check(lib.arx_groupby_minmax_init(self.minmax.data_ptr(), self.capacity, stream))
        g = int(partial["keys"].numel())
        if g == 0:
            return
        check(lib.arx_groupby_minmax_merge(self.state.data_ptr(), self.minmax.data_ptr(), self.capacity,
                                           partial["keys"].data_ptr(), partial["key_is_valid"].data_ptr(),
                                           partial["mins"].data_ptr(), partial["maxs"].data_ptr(),
                                           partial["no_nulls"].data_ptr(), g, stream))

    def finalize_min_max(self):
        """Returns (keys, key_is_valid, mins, maxs, valid) device tensors, `valid` by the options'
        skip_nulls (min_count is not consulted by the reference's min/max)."""
        lib, stream = _lib_and_stream(self.device)
        if getattr(self, "minmax", None) is None:
            raise ArrowInvalid("finalize_min_max without consume_min_max")
        g = self.num_groups()
        dev = self.device
        t = lambda dt: torch.empty(max(g, 1), dtype=dt, device=dev)  # noqa: E731
        keys, kv, sums, counts, nn = t(torch.int32), t(torch.uint8), t(torch.int64), t(torch.int64), t(torch.uint8)
        mins, maxs, valid = t(torch.int64), t(torch.int64), t(torch.uint8)
        check(lib.arx_groupby_export(self.state.data_ptr(), self.minmax.data_ptr(), keys.data_ptr(), kv.data_ptr(),
                                     sums.data_ptr(), counts.data_ptr(), nn.data_ptr(), mins.data_ptr(),
                                     maxs.data_ptr(), stream))
        check(lib.arx_groupby_minmax_finalize(mins.data_ptr(), maxs.data_ptr(), nn.data_ptr(), g,
                                              int(self.options.skip_nulls), valid.data_ptr(), stream))
        return self._typed_keys(keys[:g]), kv[:g], mins[:g], maxs[:g], valid[:g]

    def finalize_mean(self):
        """hash_mean(int64): (keys, key_is_valid, means f64, valid).  Needs the rows in BOTH consume() and
        consume_min_max() (the extrema bound the partial sums).  GroupedMeanImpl sums doubles in row order
        (hash_aggregate_numeric.cc:352-430); that equals (double)sum / count in any order exactly when every
        partial sum is an exact integer, i.e. count * max|value| < 2^53 for every group — otherwise the
        reference's own result depends on row order and this declines."""
        lib, stream = _lib_and_stream(self.device)
        if getattr(self, "minmax", None) is None:
            raise ArrowInvalid("finalize_mean without consume_min_max")
        p = self.export_min_max()
        g = int(p["keys"].numel())
        means = torch.empty(max(g, 1), dtype=torch.float64, device=self.device)
        valid = torch.empty(max(g, 1), dtype=torch.uint8, device=self.device)
        inexact = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(lib.arx_groupby_mean_i64_finalize(p["sums"].data_ptr(), p["counts"].data_ptr(), p["mins"].data_ptr(),
                                                p["maxs"].data_ptr(), p["no_nulls"].data_ptr(), g,
                                                int(self.options.skip_nulls), self.options.min_count,
                                                means.data_ptr(), valid.data_ptr(), inexact.data_ptr(), stream))
        if int(inexact.item()) != 0:
            raise ArrowNotImplementedError(
                "hash_mean(int64): a group's partial sums exceed 2^53, where the reference's row-order double "
                "accumulation is not associative (its result depends on row order); not reproducible bit for bit")
        return self._typed_keys(p["keys"]), p["key_is_valid"], means[:g], valid[:g]

    def num_groups(self) -> int:
        lib, stream = _lib_and_stream(self.device)
        n = C.c_int64(0)
        check(lib.arx_groupby_num_groups(self.state.data_ptr(), C.byref(n), stream))
        return n.value

    def export(self):
        """Partial aggregates: dict of device tensors keys(i32) key_is_valid(u8) sums(i64)
        counts(i64) no_nulls(u8), num_groups entries each, unspecified order."""
        lib, stream = _lib_and_stream(self.device)
        g = self.num_groups()
        dev = self.device
        cols = dict(keys=torch.empty(max(g, 1), dtype=torch.int32, device=dev),
                    key_is_valid=torch.empty(max(g, 1), dtype=torch.uint8, device=dev),
                    sums=torch.empty(max(g, 1), dtype=torch.int64, device=dev),
                    counts=torch.empty(max(g, 1), dtype=torch.int64, device=dev),
                    no_nulls=torch.empty(max(g, 1), dtype=torch.uint8, device=dev))
        check(lib.arx_groupby_sum_i64_export(self.state.data_ptr(), cols["keys"].data_ptr(),
                                             cols["key_is_valid"].data_ptr(), cols["sums"].data_ptr(),
                                             cols["counts"].data_ptr(), cols["no_nulls"].data_ptr(),
                                             stream))
        return {k: v[:g] for k, v in cols.items()}

    def merge(self, partial: dict) -> None:
        """Fold another state's partial aggregates in (Merge, hash_aggregate_numeric.cc:85-107)."""
        lib, stream = _lib_and_stream(self.device)
        g = int(partial["keys"].numel())
        if g == 0:
            return
        check(lib.arx_groupby_sum_i64_merge(self.state.data_ptr(), self.capacity,
                                            partial["keys"].data_ptr(), partial["key_is_valid"].data_ptr(),
                                            partial["sums"].data_ptr(), partial["counts"].data_ptr(),
                                            partial["no_nulls"].data_ptr(), g, stream))

    def finalize(self):
        """Returns (keys, key_is_valid, sums, valid) device tensors (valid: u8, 0 = null sum)."""
        lib, stream = _lib_and_stream(self.device)
        p = self.export()
        g = int(p["keys"].numel())
        valid = torch.empty(max(g, 1), dtype=torch.uint8, device=self.device)
        check(lib.arx_groupby_sum_i64_finalize(p["counts"].data_ptr(), p["no_nulls"].data_ptr(), g,
                                               int(self.options.skip_nulls), self.options.min_count,
                                               valid.data_ptr(), stream))
        return self._typed_keys(p["keys"]), p["key_is_valid"], p["sums"], valid[:g]


class RangeGroupBySum:
    """hash_sum(int64) BY int32 for keys from a narrow range, on the RANGE-PARTITIONED state of include/arrow_amd.h
    (arx_groupby_range_*; csrc/groupby_lines.h): no hash table — `partitions` dense blocks of `width` keys each,
    {uint64 sums[width] | uint64 counts[width]}.  The same operator as GroupBySum (ThreadLocalState -> Merge -> Finalize,
    acero/groupby_aggregate_node.cc:210-337; hash_aggregate_numeric.cc:44-187); rows with nulls, keys outside the
    planned range and hot keys are declined (consume returns False, nothing consumed) and go through GroupBySum."""

    SAMPLE_ROWS = 1 << 20
    MIN_ROWS = 1 << 22      # below this group_by_sum / sharded_group_by_sum do not try the state (the table operator's plans)

    @staticmethod
    def sampled_key_range(keys: Array, sample_rows: int | None = None) -> torch.Tensor:
        """Device int64[2] = {-min, max} of a sample of the key slots ({-INT32_MAX, INT32_MIN} for no rows): both fold
        over ranks with ONE all-reduce(MAX)."""
        lib, stream = _lib_and_stream(keys.device)
        pair = torch.tensor([2**31 - 1, -2**31], dtype=torch.int32, device=keys.device)
        if keys.length:
            span = keys.span()
            check(lib.arx_groupby_key_range_sampled_i32(C.byref(span), int(sample_rows or RangeGroupBySum.SAMPLE_ROWS),
                                                        pair.data_ptr(), stream))
        wide = pair.to(torch.int64)
        return torch.stack([-wide[0], wide[1]])

    @staticmethod
    def plan_for(max_rows: int, key_min: int, key_max: int, sampled: bool = True):
        """The plan for keys in [key_min, key_max] (widened by 1/64 of the range on both sides when the bounds come from a
        sample) or None when the range does not suit the state."""
        if key_min > key_max:
            return None
        pad = ((key_max - key_min + 1) // 64 + 1) if sampled else 0
        lo, hi = max(-2**31, key_min - pad), min(2**31 - 1, key_max + pad)
        plan = _lib.ArxRangePlan()
        rc = _lib.get_lib().arx_groupby_range_plan(int(max_rows), lo, hi, C.byref(plan))
        if rc == _lib.ARX_NOT_IMPLEMENTED:
            return None
        check(rc)
        return plan

    def __init__(self, plan, device=None, options: ScalarAggregateOptions | None = None):
        from .array import default_device

        self.plan = plan
        self.device = torch.device(device) if device is not None else default_device()
        self.options = options or ScalarAggregateOptions()
        self.state = torch.zeros(int(plan.state_bytes) // 8, dtype=torch.int64, device=self.device)

    def partition_bytes(self) -> int:
        return int(self.plan.width) * 16

    def consume(self, keys: Array, values: Array) -> bool:
        """Adds the rows' groups; False = declined (nothing consumed): nulls, a key outside the plan, a hot key."""
        if keys.type != int32 or values.type != int64:
            return False
        if (keys.null_count != 0 and keys.buffers[0] is not None) or (values.null_count != 0 and values.buffers[0] is not None):
            return False
        if keys.length == 0:
            return True
        lib, stream = _lib_and_stream(self.device)
        probe = _lib.ArxRangePlan()
        check(lib.arx_groupby_range_plan(keys.length, self.plan.key_min, self.plan.key_min + int(self.plan.slots) - 1, C.byref(probe)))
        ws = _workspace(self.device, int(probe.workspace_bytes) + 256, "groupby")
        ks, vs = keys.span(), values.span()
        rc = lib.arx_groupby_range_sum_i64_consume(self.state.data_ptr(), C.byref(self.plan), C.byref(ks), C.byref(vs),
                                                   ws.data_ptr(), ws.numel(), stream)
        if rc in (_lib.ARX_CAPACITY_ERROR, _lib.ARX_NOT_IMPLEMENTED):
            return False
        check(rc)
        return True

    def merge_blocks(self, first_partition: int, num_partitions: int, others: torch.Tensor, num_others: int) -> None:
        """partitions [first, first + num) += the same partitions of num_others states lying one behind the other in
        `others` (a uint8 / int64 tensor of num_others x num_partitions x width x 16 bytes)."""
        if num_partitions == 0 or num_others == 0:
            return
        lib, stream = _lib_and_stream(self.device)
        stride = num_partitions * self.partition_bytes()
        dst = self.state.data_ptr() + first_partition * self.partition_bytes()
        check(lib.arx_groupby_range_merge(dst, others.data_ptr(), self.plan.width, num_partitions, num_others, stride, stream))

    def finalize(self, first_partition: int = 0, num_partitions: int | None = None, blocks: torch.Tensor | None = None):
        """(keys, key_is_valid, sums, valid) of the partitions [first, first + num), ascending by key — the tuple
        GroupBySum.finalize returns.  `blocks`: finalize these partitions' blocks from another buffer instead of the state."""
        lib, stream = _lib_and_stream(self.device)
        if num_partitions is None:
            num_partitions = int(self.plan.partitions) - first_partition
        slots = num_partitions * int(self.plan.width)
        dev = self.device
        keys = torch.empty(max(slots, 1), dtype=torch.int32, device=dev)
        sums = torch.empty(max(slots, 1), dtype=torch.int64, device=dev)
        valid = torch.empty(max(slots, 1), dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = alloc(lib.arx_groupby_range_finalize_workspace_bytes(slots), dev)
        src = (blocks.data_ptr() if blocks is not None else self.state.data_ptr() + first_partition * self.partition_bytes())
        check(lib.arx_groupby_range_finalize(src, self.plan.key_min + first_partition * int(self.plan.width), self.plan.width,
                                             num_partitions, int(self.options.min_count), ws.data_ptr(), ws.numel(),
                                             keys.data_ptr(), sums.data_ptr(), None, valid.data_ptr(), count.data_ptr(), stream))
        g = int(count.item())
        return keys[:g], torch.ones(g, dtype=torch.uint8, device=dev), sums[:g], valid[:g]


def indices_nonzero(arr: Array) -> Array:
    """compute::IndicesNonZero (kernels/vector_selection.cc:352, vector_selection.cc DoNonZero): the uint64 positions
    of the elements that are valid and non-zero (true for booleans), ascending.  = GetTakeIndices of the mask
    `arr != 0` (nulls dropped), widened to uint64."""
    if arr.type == bool_:
        mask = arr
    elif arr.type.name in _NUM_TYPE_ID:
        wide = arr
        if arr.type not in (int64, float64):
            wide = cast(arr, float64 if arr.type.name in ("float", "double") else
                        (uint64 if arr.type.name.startswith("uint") else int64))
            if wide.type == uint64:       # same zero-ness, same bits
                wide = Array(int64, wide.length, wide.buffers, wide.null_count, wide.offset)
        mask = call_function("not_equal", [wide, _wrap_scalar(0 if wide.type == int64 else 0.0, wide)])
    else:
        raise ArrowNotImplementedError(f"indices_nonzero: {arr.type.name} values")
    idx = get_take_indices(mask)
    return cast(idx, uint64)


def group_by_mean(keys: Array, values: Array, capacity: int | None = None,
                  options: ScalarAggregateOptions | None = None):
    """Table.group_by(k).aggregate([(v, 'mean')]) for one int32 key and one int64 value (see finalize_mean)."""
    cap = capacity or max(16, 2 * keys.length + 2)
    op = GroupBySum(cap, keys.device, options)
    op.consume(keys, values)
    op.consume_min_max(keys, values)
    return op.finalize_mean()


def _groups_by_first_row(arr: Array, capacity: int | None, with_counts: bool):
    """The distinct int32 values of `arr` ordered by first appearance, from the fused group-by table:
    the rows' own numbers go through the min reducer (first row of every group) and, for
    value_counts, through the sum consume (whose `counts` column is the number of rows per group);
    the G groups are then ordered by first row with array_sort_indices + take.  Everything
    row-sized runs in the HIP kernels; torch only numbers the rows.
    Returns (keys int32 tensor[G], key_is_valid uint8 tensor[G], counts Array | None)."""
    if arr.type != int32:
        raise ArrowNotImplementedError("unique / value_counts / dictionary_encode on the gfx950 path: int32 values only")
    dev = arr.device
    n = arr.length
    rows = Array(int64, n, [None, torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8)], 0, 0)
    op = GroupBySum(capacity or max(16, 2 * n + 2), dev)
    if with_counts:
        op.consume(arr, rows)
    op.consume_min_max(arr, rows)
    p = op.export_min_max()
    g = int(p["keys"].numel())
    as_arr = lambda t, col: Array(t, g, [None, col.contiguous().view(torch.uint8)], 0, 0)  # noqa: E731
    order = call_function("array_sort_indices", [as_arr(int64, p["mins"])], ArraySortOptions())
    keys = take(as_arr(int32, p["keys"]), order, boundscheck=False).data[: g * 4].view(torch.int32)
    kvalid = take(as_arr(uint8, p["key_is_valid"]), order, boundscheck=False).data[:g]
    counts = take(as_arr(int64, p["counts"]), order, boundscheck=False) if with_counts else None
    return keys, kvalid, counts


def _values_array(keys: torch.Tensor, kvalid: torch.Tensor) -> Array:
    """int32 Array over `keys` whose only possible null (all nulls are one group) is cleared in a fresh bitmap."""
    g = int(keys.numel())
    dev = keys.device
    validity, null_count = None, 0
    null_pos = torch.nonzero(kvalid == 0)
    if null_pos.numel():
        pos = int(null_pos[0])
        validity = torch.full((bitmap_nbytes(g),), 0xFF, dtype=torch.uint8, device=dev)
        validity[g // 8:] = 0
        if g % 8:
            validity[g // 8] = (1 << (g % 8)) - 1
        validity[pos // 8] = int(validity[pos // 8]) & ~(1 << (pos % 8))
        null_count = 1
    data = keys.contiguous().view(torch.uint8)
    if data.numel() < 64:      # keep the 64-byte padding every device buffer has
        padded = alloc(64, dev, zero=True)
        padded[: data.numel()] = data
        data = padded
    return Array(int32, g, [validity, data], null_count, 0)


def _first_occurrence_groups(arr: Array, capacity: int | None, with_counts: bool):
    dev = arr.device
    if arr.length == 0:
        empty = Array(int32, 0, [None, alloc(0, dev)], 0, 0)
        return empty, Array(int64, 0, [None, alloc(0, dev)], 0, 0)
    keys, kvalid, counts = _groups_by_first_row(arr, capacity, with_counts)
    return _values_array(keys, kvalid), counts


def dictionary_encode(arr: Array, null_encoding: str = "mask", capacity: int | None = None):
    """compute::DictionaryEncode (DictEncodeAction, kernels/vector_hash.cc:173-270) for int32:
    (indices int32 Array, dictionary int32 Array); the dictionary holds the distinct values in order
    of first appearance.  null_encoding "mask" (default): nulls stay null in the indices and are not
    in the dictionary; "encode": the null is a dictionary entry like any value.
    Groups -> first rows -> order (as unique), the positions merged back into a table as its "sums",
    then one read-only lookup per row (arx_groupby_lookup_i32)."""
    if null_encoding not in ("mask", "encode"):
        raise ArrowInvalid("null_encoding must be 'mask' or 'encode'")
    dev = arr.device
    n = arr.length
    if n == 0:
        return (Array(int32, 0, [None, alloc(0, dev)], 0, 0), Array(int32, 0, [None, alloc(0, dev)], 0, 0))
    keys, kvalid, _ = _groups_by_first_row(arr, capacity, False)
    if null_encoding == "mask":
        keep = kvalid != 0
        keys, kvalid = keys[keep].contiguous(), kvalid[keep].contiguous()
    g = int(keys.numel())
    lib, stream = _lib_and_stream(dev)
    table = GroupBySum(max(16, 2 * g + 2), dev)
    if g:
        ids = torch.arange(g, dtype=torch.int64, device=dev)
        zeros = torch.zeros(g, dtype=torch.int64, device=dev)
        check(lib.arx_groupby_sum_i64_merge(table.state.data_ptr(), table.capacity, keys.data_ptr(), kvalid.data_ptr(),
                                            ids.data_ptr(), zeros.data_ptr(), None, g, stream))
    out = alloc(n * 4, dev)
    sp = arr.span()
    check(lib.arx_groupby_lookup_i32(table.state.data_ptr(), table.capacity, C.byref(sp), out.data_ptr(), stream))
    if null_encoding == "mask":
        validity, nc = _propagate_validity([arr], n, dev)
    else:
        validity, nc = None, 0
    return Array(int32, n, [validity, out], nc, 0), _values_array(keys, kvalid)


def unique(arr: Array, capacity: int | None = None) -> Array:
    """compute::Unique (api_vector.cc; UniqueAction, kernels/vector_hash.cc:65-120) for int32:
    distinct values in order of first appearance, the nulls as one entry.  `capacity`: slots of
    the device hash table (default 2 * length + 2; pass ~2x the expected distinct count to save HBM)."""
    return _first_occurrence_groups(arr, capacity, False)[0]


def value_counts(arr: Array, capacity: int | None = None):
    """compute::ValueCounts (ValueCountsAction, kernels/vector_hash.cc:125-190) for int32:
    (values, counts int64) in order of first appearance; nulls are counted as one value."""
    return _first_occurrence_groups(arr, capacity, True)


class _GrouperLevel:
    """One device table of csrc/grouper.hip: key rows of up to 8 fixed-width columns and 16 bytes (`Grouper` below
    chains several of these for wider rows)."""

    def __init__(self, key_types, max_groups: int, device=None):
        from .array import default_device

        self.key_types = list(key_types)
        self.device = torch.device(device) if device is not None else default_device()
        self.max_groups = max(1, int(max_groups))
        self._widths = (C.c_int32 * len(self.key_types))(*[t.byte_width for t in self.key_types])
        lib, stream = _lib_and_stream(self.device)
        self.state = alloc(lib.arx_grouper_state_bytes(self.max_groups) + 256, self.device)
        self._state_ptr = (self.state.data_ptr() + 255) & ~255
        check(lib.arx_grouper_init(self._state_ptr, self.max_groups, stream))

    def reset(self) -> None:
        lib, stream = _lib_and_stream(self.device)
        check(lib.arx_grouper_init(self._state_ptr, self.max_groups, stream))

    def _spans(self, batch):
        cols = list(batch.values) if isinstance(batch, ExecBatch) else list(batch)
        if len(cols) != len(self.key_types):
            raise ArrowInvalid(f"Grouper: expected {len(self.key_types)} key columns, got {len(cols)}")
        n = cols[0].length
        for c, t in zip(cols, self.key_types):
            if c.type != t:
                raise ArrowInvalid(f"Grouper: key column of type {c.type.name}, expected {t.name}")
            if c.length != n:
                raise ArrowInvalid("Array arguments must all be the same length")
        spans = (_lib.ArxSpan * len(cols))(*[c.span() for c in cols])
        return cols, spans, n

    def _run(self, batch, lookup: bool):
        lib, stream = _lib_and_stream(self.device)
        cols, spans, n = self._spans(batch)
        ids = alloc(max(n, 1) * 4, self.device)
        ws_bytes = lib.arx_grouper_consume_workspace_bytes(n)
        ws = _workspace(self.device, ws_bytes + 256, "grouper")
        ws_ptr = (ws.data_ptr() + 255) & ~255
        ws_len = ws.numel() - (ws_ptr - ws.data_ptr())
        if lookup:
            valid = alloc(bitmap_nbytes(n), self.device, zero=True)
            check(lib.arx_grouper_lookup(self._state_ptr, self.max_groups, spans, self._widths, len(cols), ws_ptr,
                                         ws_len, ids.data_ptr(), valid.data_ptr(), stream))
            return Array(uint32, n, [valid, ids], kUnknownNullCount, 0)
        check(lib.arx_grouper_consume(self._state_ptr, self.max_groups, spans, self._widths, len(cols), ws_ptr, ws_len,
                                      ids.data_ptr(), stream))
        return Array(uint32, n, [None, ids], 0, 0)

    def consume(self, batch) -> Array:
        return self._run(batch, lookup=False)

    def lookup(self, batch) -> Array:
        return self._run(batch, lookup=True)

    def populate(self, batch) -> None:
        self._run(batch, lookup=False)

    @property
    def num_groups(self) -> int:
        lib, stream = _lib_and_stream(self.device)
        n = C.c_int64(0)
        check(lib.arx_grouper_num_groups(self._state_ptr, C.byref(n), stream))
        return n.value

    def get_uniques(self):
        """The unique key rows as one Array per key column, in group-id order."""
        lib, stream = _lib_and_stream(self.device)
        g = self.num_groups
        out = []
        for j, t in enumerate(self.key_types):
            data = alloc(max(g, 1) * t.byte_width, self.device)
            valid = alloc(bitmap_nbytes(g), self.device, zero=True)
            nulls = C.c_int64(0)
            check(lib.arx_grouper_get_uniques(self._state_ptr, self.max_groups, self._widths, len(self.key_types), j,
                                              data.data_ptr(), valid.data_ptr(), C.byref(nulls), stream))
            out.append(Array(t, g, [valid if nulls.value else None, data], nulls.value, 0))
        return ExecBatch(out, g)


kGrouperLevelBytes = 16   # csrc/grouper.hip: one table holds key rows of up to 16 bytes ...
kGrouperLevelKeys = 8     # ... and 8 columns


def _grouper_levels(widths):
    """Key columns -> the levels of the chain: level 0 takes columns while they fit 16 bytes / 8 columns, every later
    level takes the previous level's uint32 group id (4 bytes, never null) plus the columns that fit beside it."""
    levels, cur, used = [], [], 0
    for j, w in enumerate(widths):
        if cur and (used + w > kGrouperLevelBytes or len(cur) + (1 if levels else 0) >= kGrouperLevelKeys):
            levels.append(cur)
            cur, used = [], 4
        cur.append(j)
        used += w
    levels.append(cur)
    return levels


class Grouper:
    """arrow::compute::Grouper (compute/row/grouper.h:104-137) over key columns that live in HBM: `consume(batch)`
    -> uint32 group ids (:121), `lookup` (:126; unseen keys -> null), `populate` (:130), `get_uniques` (:134),
    `num_groups` (:137), `reset` (:115).  Key rows = one or several fixed-width columns (byte widths 1, 2, 4, 8),
    compared by their bits with null as a key value of its own, as the reference's row encoder does (GrouperFastImpl,
    row/grouper.cc:555-973; any number of fixed-width columns, :559-611).  The k-th distinct key row in row order gets
    id k.  `max_groups` bounds the distinct rows over the Grouper's life.

    One device table holds rows of up to 16 bytes (32-byte slots, csrc/grouper.hip).  Wider rows go through a CHAIN of
    tables instead of wider slots: level 0 maps the first columns to ids, level s maps (id of level s-1, next columns) —
    a 4-byte id stands for everything to its left, so every slot stays 32 bytes whatever the row width, and the ids of
    the last level are the ids of the whole row, in order of first appearance (a row's tuple (prefix id, columns)
    appears first exactly where the row does).  GetUniques walks back: level s keeps its groups' level s-1 ids, so the
    columns of level s are gathered through the composed id maps (`take`)."""

    def __init__(self, key_types, max_groups: int, device=None):
        from .array import default_device

        self.key_types = list(key_types)
        if not 1 <= len(self.key_types) <= 32:
            raise ArrowNotImplementedError(f"Grouper: 1 to 32 key columns (got {len(self.key_types)})")
        for t in self.key_types:
            if t.bit_width not in (8, 16, 32, 64):
                raise ArrowNotImplementedError(f"Grouper: keys of type {t.name}")
        self.device = torch.device(device) if device is not None else default_device()
        self.max_groups = max(1, int(max_groups))
        self._level_cols = _grouper_levels([t.byte_width for t in self.key_types])
        self._levels = []
        for s, cols in enumerate(self._level_cols):
            types = ([uint32] if s else []) + [self.key_types[j] for j in cols]
            self._levels.append(_GrouperLevel(types, self.max_groups, self.device))

    @classmethod
    def make(cls, key_types, max_groups: int, device=None) -> "Grouper":
        """Grouper::Make (:110)."""
        return cls(key_types, max_groups, device)

    @property
    def num_levels(self) -> int:
        return len(self._levels)

    def reset(self) -> None:
        for lv in self._levels:
            lv.reset()

    def _check(self, batch):
        cols = list(batch.values) if isinstance(batch, ExecBatch) else list(batch)
        if len(cols) != len(self.key_types):
            raise ArrowInvalid(f"Grouper: expected {len(self.key_types)} key columns, got {len(cols)}")
        n = cols[0].length
        for c, t in zip(cols, self.key_types):
            if c.type != t:
                raise ArrowInvalid(f"Grouper: key column of type {c.type.name}, expected {t.name}")
            if c.length != n:
                raise ArrowInvalid("Array arguments must all be the same length")
        return cols

    def _run(self, batch, lookup: bool):
        cols = self._check(batch)
        ids = None
        for lv, idx in zip(self._levels, self._level_cols):
            # a Lookup's unseen prefix is a null id: no consumed row has one, so the next level does not find it either
            part = ([ids] if ids is not None else []) + [cols[j] for j in idx]
            ids = lv._run(part, lookup)
        return ids

    def consume(self, batch) -> Array:
        return self._run(batch, lookup=False)

    def lookup(self, batch) -> Array:
        return self._run(batch, lookup=True)

    def populate(self, batch) -> None:
        self._run(batch, lookup=False)

    @property
    def num_groups(self) -> int:
        return self._levels[-1].num_groups

    def get_uniques(self):
        """The unique key rows as one Array per key column, in group-id order."""
        out = [None] * len(self.key_types)
        to_level = None   # final group id -> group id of the level at hand (None: the identity, at the last level)
        for s in range(len(self._levels) - 1, -1, -1):
            u = self._levels[s].get_uniques().values
            own = u[1:] if s else u
            for j, a in zip(self._level_cols[s], own):
                out[j] = a if to_level is None else take(a, to_level, boundscheck=False)
            if s:
                to_level = u[0] if to_level is None else take(u[0], to_level, boundscheck=False)
        return ExecBatch(out, self.num_groups)



def binary_key_columns(arr: Array):
    """A utf8 / binary KEY column as the fixed-width virtual key columns the chain of Grouper tables takes
    (arx_binary_key_lengths / arx_binary_key_chunk; the var-length part of the reference's key rows,
    row/grouper.cc:559-611): (lengths uint32 Array — 0xFFFFFFFF for a null, [(lo uint64 Array, hi uint32 Array) per
    12-byte chunk, zero-padded past the string's end])."""
    lib, stream = _lib_and_stream(arr.device)
    n = arr.length
    bspan = arr.binary_span()
    lens = alloc(max(n, 1) * 4, arr.device)
    ws = _workspace(arr.device, 256, "binary_key")
    max_len = C.c_int64(0)
    check(lib.arx_binary_key_lengths(C.byref(bspan), lens.data_ptr(), C.byref(max_len), ws.data_ptr(), stream))
    chunks = []
    for c in range((max_len.value + 11) // 12):
        lo, hi = alloc(max(n, 1) * 8, arr.device), alloc(max(n, 1) * 4, arr.device)
        check(lib.arx_binary_key_chunk(C.byref(bspan), c, lo.data_ptr(), hi.data_ptr(), stream))
        chunks.append((Array(uint64, n, [None, lo], 0, 0), Array(uint32, n, [None, hi], 0, 0)))
    return Array(uint32, n, [None, lens], 0, 0), chunks


def group_first_rows(group_ids: Array, num_groups: int) -> Array:
    """first_rows[g] = the smallest row whose group id is g (arx_group_first_rows): `take(keys, first_rows)` are the
    unique key rows in group-id order — how var-width keys come back out of the chain of tables."""
    lib, stream = _lib_and_stream(group_ids.device)
    out = alloc(max(num_groups, 1) * 4, group_ids.device)
    check(lib.arx_group_first_rows(group_ids.buffers[1].data_ptr() + 4 * group_ids.offset, group_ids.length, num_groups,
                                   out.data_ptr(), stream))
    return Array(uint32, num_groups, [None, out], 0, 0)


_GROUP_BY_AGGREGATES = ("hash_sum", "hash_count", "hash_mean")


def group_by(keys, aggregates, max_groups: int | None = None, options: ScalarAggregateOptions | None = None):
    """What GroupByNode does with one batch (acero/groupby_aggregate_node.cc:210-337): Grouper::Consume over the key
    columns, then every aggregate's {resize, consume} on (values, group ids), then finalize + GetUniques.
    keys: list of Arrays (any fixed-width types up to 16 bytes per row — int64 keys, several key columns);
    aggregates: list of (values Array, "hash_sum" | "hash_count" | "hash_mean") over int64 values.
    Returns (unique key columns, [one result Array per aggregate]), rows in group-id order (first appearance)."""
    keys = list(keys)
    n = keys[0].length
    g = Grouper([k.type for k in keys], max_groups or max(16, n), keys[0].device)
    ids = g.consume(keys)
    num_groups = g.num_groups
    results = []
    states = {}
    for values, fn in aggregates:
        if fn not in _GROUP_BY_AGGREGATES:
            raise ArrowNotImplementedError(f"group_by: aggregate {fn}")
        if id(values) not in states:   # sum / count / mean of one column share its state
            st = _hash_sum_init(options, keys[0].device)
            _hash_sum_resize(st, num_groups)
            _hash_sum_consume(st, [GroupBySum._normalise_value(GroupBySum.__new__(GroupBySum), values), ids])
            states[id(values)] = st
        st = states[id(values)]
        if fn == "hash_sum":
            out = _hash_sum_finalize(st)
            if values.type.name.startswith("uint"):
                out = Array(uint64, out.length, out.buffers, out.null_count, out.offset)
        elif fn == "hash_count":
            out = Array(int64, num_groups, [None, st.counts[:num_groups].contiguous().view(torch.uint8)], 0, 0)
        else:
            out = _hash_mean_from_dense(st, values)
        results.append(out)
    return g.get_uniques(), results


def _hash_mean_from_dense(st: "GroupedSumInt64State", values: Array) -> Array:
    """hash_mean(int64) from the dense sums / counts: GroupedMeanImpl::Finalize (hash_aggregate_numeric.cc:352-430)
    divides a sum that was accumulated as DOUBLES in row order; that equals double(exact sum) / count exactly when
    every partial sum is an integer below 2^53 — guaranteed here when count * max|value| < 2^53 for every group,
    which is checked with the column's extrema (a bound, not per group); otherwise declined."""
    if values.type != int64:
        raise ArrowNotImplementedError(f"group_by: hash_mean of {values.type.name}")
    g = st.num_groups
    mm = min_max(values)
    lo, hi = mm if mm else (0, 0)
    bound = builtins.max(abs(int(lo)), abs(int(hi)))
    lib, stream = _lib_and_stream(st.device)
    bits = alloc(bitmap_nbytes(g), st.device, zero=True)
    counter = torch.zeros(8, dtype=torch.uint8, device=st.device)
    check(lib.arx_hash_sum_i64_finalize(st.counts.data_ptr(), st.null_seen.data_ptr(), g, int(st.options.skip_nulls),
                                        st.options.min_count, bits.data_ptr(), counter.data_ptr(), stream))
    means = alloc(max(g, 1) * 8, st.device)
    inexact = torch.zeros(1, dtype=torch.int32, device=st.device)
    check(lib.arx_hash_mean_i64_finalize(st.sums.data_ptr(), st.counts.data_ptr(), g, bound, means.data_ptr(),
                                         inexact.data_ptr(), stream))
    if int(inexact.item()) != 0:
        raise ArrowNotImplementedError(
            "hash_mean(int64): a group's partial sums may exceed 2^53, where the reference's row-order double "
            "accumulation is not associative (its result depends on row order); not reproducible bit for bit")
    nulls = g - int(counter.cpu().view(torch.int64)[0])
    return Array(float64, g, [bits if nulls else None, means], nulls, 0)


def group_by_sum(keys: Array, values: Array, capacity: int | None = None,
                 options: ScalarAggregateOptions | None = None):
    """Table.group_by(k).aggregate([(v, 'sum')]) for one int32 key and one int64 value.  Large null-free inputs whose keys
    come from a narrow range go through the range-partitioned state (RangeGroupBySum: no hash table, groups in key order);
    everything else — and whatever that state declines — through the table operator."""
    if (keys.type == int32 and values.type == int64 and keys.length >= RangeGroupBySum.MIN_ROWS and
            not (keys.null_count != 0 and keys.buffers[0] is not None) and not (values.null_count != 0 and values.buffers[0] is not None)):
        neg_lo, hi = [int(x) for x in RangeGroupBySum.sampled_key_range(keys).cpu().tolist()]
        plan = RangeGroupBySum.plan_for(keys.length, -neg_lo, hi) if -neg_lo <= hi else None
        if plan is not None:
            st = RangeGroupBySum(plan, keys.device, options)
            if st.consume(keys, values):
                return st.finalize()
    cap = capacity or max(16, 2 * keys.length + 2)
    op = GroupBySum(cap, keys.device, options)
    op.consume(keys, values)
    return op.finalize()
