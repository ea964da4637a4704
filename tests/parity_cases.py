"""Parity checks shared by the GPU tests (real library, -m gpu) and the emulator tests
(kernel sources under the CPU SIMT emulator).  Every check runs the op through
arrow_amd.compute -> C ABI and compares with the C oracle (bit-exact) and, when the wheel
is importable, with the reference's own build (pyarrow)."""
from __future__ import annotations

import os

import numpy as np
import pytest

from oracle import oracle as O

from . import util
from .util import HostArray, assert_equal, device_bitmap_to_bool, oracle_bitmap_to_bool, pa, pc


def _logical_valid(arr):
    if arr.validity is None:
        return np.ones(arr.length, dtype=bool), True
    bits, pad_ok = device_bitmap_to_bool(arr.validity, arr.length)
    return bits, pad_ok


def _data_np(arr, dtype):
    raw = arr.data.cpu().numpy()
    return raw[: arr.length * np.dtype(dtype).itemsize].view(dtype)


# ------------------------------------------------------------------ filter
def check_filter(amd, values: HostArray, mask: HostArray, null_selection: str, use_pyarrow=True):
    code = 1 if null_selection == "emit_null" else 0
    dv, dm = values.to_device(amd), mask.to_device(amd)
    out = amd.compute.filter(dv, dm, null_selection)
    want_vals, want_bm = O.filter(values.data_bytes(), values.valid_bitmap(), values.offset,
                                  mask.data_bytes(), mask.valid_bitmap(), mask.offset, mask.length,
                                  code, True)
    tag = f"filter[{values.dtype},n={values.length},{null_selection},voff={values.offset},moff={mask.offset}]"
    assert out.length == len(want_vals), f"{tag}: length {out.length} vs {len(want_vals)}"
    assert out.offset == 0
    assert_equal(_data_np(out, values.dtype), want_vals, tag + " data bytes")
    got_valid, pad_ok = _logical_valid(out)
    assert_equal(got_valid, oracle_bitmap_to_bool(want_bm, out.length), tag + " validity")
    assert pad_ok, tag + ": padding bits of the validity bitmap are not zero"
    # allocate_validity rule + null_count bookkeeping (vector_selection_filter_internal.cc:462-472)
    if dv.null_count == 0 and dm.null_count == 0:
        assert out.validity is None and out.null_count == 0
    # arx_filter_count_nulls: the output's length AND null count from the counting pass alone (what the plugin puts into
    # the device-resident output instead of kUnknownNullCount)
    import ctypes as C

    from arrow_amd import _lib as L
    from arrow_amd.array import alloc, current_stream

    lib = L.get_lib()
    ws = alloc(lib.arx_filter_workspace_bytes(mask.length) + 64, dv.device)
    ws_ptr = (ws.data_ptr() + 63) & ~63
    vspan, mspan = dv.span(), dm.span()
    n_out, n_null = C.c_int64(-1), C.c_int64(-1)
    L.check(lib.arx_filter_count_nulls(C.byref(vspan), C.byref(mspan), code, ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()),
                                       C.byref(n_out), C.byref(n_null), current_stream(dv.device)))
    assert n_out.value == out.length and n_null.value == int(out.length - got_valid.sum()), (tag, n_out.value, n_null.value, out.length, int(got_valid.sum()))
    if use_pyarrow and pc is not None:
        ref = pc.filter(values.to_pyarrow(), mask.to_pyarrow(), null_selection_behavior=null_selection)
        assert len(ref) == out.length
        rvalid = ~np.asarray(ref.is_null())
        assert_equal(got_valid, rvalid, tag + " validity vs pyarrow")
        rvals = ref.fill_null(0).to_numpy(zero_copy_only=False)
        assert_equal(_data_np(out, values.dtype)[got_valid], rvals[rvalid], tag + " values vs pyarrow")
    return out


def check_mask_to_indices(amd, mask: HostArray, null_selection: str):
    code = 1 if null_selection == "emit_null" else 0
    dm = mask.to_device(amd)
    out = amd.compute.get_take_indices(dm, null_selection)
    want, want_bm = O.mask_to_indices(mask.data_bytes(), mask.valid_bitmap(), mask.offset, mask.length,
                                      code, True)
    tag = f"get_take_indices[n={mask.length},{null_selection},off={mask.offset}]"
    assert out.length == len(want), tag
    assert out.type.np_dtype == want.dtype, f"{tag}: index type {out.type.name} vs {want.dtype}"
    assert_equal(_data_np(out, want.dtype), want, tag + " indices")
    got_valid, pad_ok = _logical_valid(out)
    assert_equal(got_valid, oracle_bitmap_to_bool(want_bm, out.length), tag + " validity")
    assert pad_ok
    return out


# ------------------------------------------------------------------ take
def check_take(amd, values: HostArray, indices: HostArray, boundscheck=True, use_pyarrow=True):
    dv, di = values.to_device(amd), indices.to_device(amd)
    out = amd.compute.take(dv, di, boundscheck=boundscheck)
    want, want_bm, want_vc = O.take(values.data_bytes(), values.valid_bitmap(), values.offset,
                                    np.ascontiguousarray(indices.values), indices.valid_bitmap(),
                                    indices.offset, indices.length, True)
    tag = f"take[{values.dtype},idx={indices.dtype},m={indices.length},voff={values.offset},ioff={indices.offset}]"
    assert out.length == indices.length
    assert_equal(_data_np(out, values.dtype), want, tag + " data bytes")
    got_valid, pad_ok = _logical_valid(out)
    assert_equal(got_valid, oracle_bitmap_to_bool(want_bm, out.length), tag + " validity")
    assert pad_ok
    assert out.null_count == indices.length - want_vc, tag + " null_count"
    if use_pyarrow and pc is not None:
        ref = pc.take(values.to_pyarrow(), indices.to_pyarrow(), boundscheck=boundscheck)
        rvalid = ~np.asarray(ref.is_null())
        assert_equal(got_valid, rvalid, tag + " validity vs pyarrow")
        assert ref.null_count == out.null_count
        rvals = ref.fill_null(0).to_numpy(zero_copy_only=False)
        assert_equal(_data_np(out, values.dtype)[got_valid], rvals[rvalid], tag + " values vs pyarrow")
    return out


def check_take_record_batch(amd, rng, dtypes, n, m, idx_dtype=np.uint32, value_null_p=0.1, index_null_p=0.05,
                            offsets=True, use_pyarrow=True):
    """`take` of a RecordBatch (TakeRAR, vector_selection_take_internal.cc:619-633): the fixed-width columns go through
    arx_take_columns (one launch); every column must equal the oracle's / pyarrow's per-column take — values, validity,
    padding bits, null counts — including columns with and without nulls, different offsets, and an out-of-bounds index
    reported once with the reference's text."""
    cols, host = {}, {}
    for j, dt in enumerate(dtypes):
        h = util.random_array(rng, dt, n, null_p=(value_null_p if j % 3 != 1 else 0.0), offset=(j * 3 if offsets else 0))
        host[f"c{j}"] = h
        cols[f"c{j}"] = h.to_device(amd)
    idx = util.random_array(rng, idx_dtype, m, null_p=index_null_p, offset=(2 if offsets else 0), lo=0, hi=max(n - 1, 0))
    di = idx.to_device(amd)
    out = amd.compute.take(amd.compute.RecordBatch(cols), di)
    assert list(out.columns) == list(cols)
    for name, h in host.items():
        got = out.columns[name]
        want, want_bm, want_vc = O.take(h.data_bytes(), h.valid_bitmap(), h.offset, np.ascontiguousarray(idx.values),
                                        idx.valid_bitmap(), idx.offset, m, True)
        tag = f"take_record_batch[{name}:{h.dtype},n={n},m={m}]"
        assert got.length == m
        assert_equal(_data_np(got, h.dtype), want, tag + " data bytes")
        gv, pad_ok = _logical_valid(got)
        assert pad_ok
        assert_equal(gv, oracle_bitmap_to_bool(want_bm, m), tag + " validity")
        assert got.null_count == m - want_vc, tag + " null_count"
        if use_pyarrow and pc is not None:
            ref = pc.take(h.to_pyarrow(), idx.to_pyarrow())
            assert ref.null_count == got.null_count
            rvalid = ~np.asarray(ref.is_null())
            assert_equal(gv, rvalid, tag + " validity vs pyarrow")
    if m:
        bad = idx.values.copy()
        pos = idx.offset + m // 2
        bad[pos] = n + 7
        valid2 = None
        if idx.valid is not None:   # make sure the offending slot is a valid index
            valid2 = idx.valid.copy()
            valid2[pos] = True
        idx2 = HostArray(bad, valid2, idx.offset, m)
        with pytest.raises(amd.ArrowIndexError) as ei:
            amd.compute.take(amd.compute.RecordBatch(cols), idx2.to_device(amd))
        assert str(ei.value) == f"Index {n + 7} out of bounds", str(ei.value)


def check_take_out_of_bounds(amd, values: HostArray, indices: HostArray):
    dv, di = values.to_device(amd), indices.to_device(amd)
    bad = O.check_index_bounds(np.ascontiguousarray(indices.values), indices.valid_bitmap(),
                               indices.offset, indices.length, values.length)
    assert bad is not None, "test case must contain an out-of-bounds index"
    with pytest.raises(amd.ArrowIndexError) as ei:
        amd.compute.take(dv, di)
    assert str(ei.value) == f"Index {bad} out of bounds", str(ei.value)
    if pc is not None:
        with pytest.raises(pa.lib.ArrowIndexError) as ri:
            pc.take(values.to_pyarrow(), indices.to_pyarrow())
        assert str(ri.value) == str(ei.value)


# ------------------------------------------------------------------ binary / utf8 take + filter
def _binary_out(out):
    offs = out.buffers[1].cpu().numpy()[: (out.length + 1) * 4].view(np.int32)
    total = int(offs[-1])
    data = out.buffers[2].cpu().numpy()[:total]
    return offs, data


def _check_binary_result(out, want, tag, ref):
    want_off, want_data, want_bm, want_vc = want
    assert out.offset == 0
    offs, data = _binary_out(out)
    assert_equal(offs, want_off, tag + " offsets")
    assert_equal(data, want_data, tag + " data bytes")
    got_valid, pad_ok = _logical_valid(out)
    assert_equal(got_valid, oracle_bitmap_to_bool(want_bm, out.length), tag + " validity")
    assert pad_ok, tag + ": padding bits of the validity bitmap are not zero"
    assert out.null_count == out.length - want_vc, tag + " null_count"
    if ref is not None:
        assert out.to_pyarrow().equals(ref), tag + " vs pyarrow"
        # the reference's own buffers: offsets start at 0 and are dense, like ours
        roffs = np.frombuffer(ref.buffers()[1], dtype=np.int32)[ref.offset: ref.offset + len(ref) + 1]
        assert_equal(offs - offs[0], roffs - roffs[0], tag + " offsets vs pyarrow")


def check_binary_take(amd, values, indices: HostArray, boundscheck=True, use_pyarrow=True):
    dv, di = values.to_device(amd), indices.to_device(amd)
    out = amd.compute.take(dv, di, boundscheck=boundscheck)
    want = O.binary_take(values.offsets, values.data, values.valid_bitmap(), values.offset,
                         np.ascontiguousarray(indices.values), indices.valid_bitmap(), indices.offset,
                         indices.length)
    tag = f"binary_take[idx={indices.dtype},m={indices.length},voff={values.offset},ioff={indices.offset}]"
    assert out.length == indices.length and out.type == dv.type
    ref = None
    if use_pyarrow and pc is not None:
        ref = pc.take(values.to_pyarrow(), indices.to_pyarrow(), boundscheck=boundscheck)
    _check_binary_result(out, want, tag, ref)
    return out


def check_binary_filter(amd, values, mask: HostArray, null_selection: str, use_pyarrow=True):
    code = 1 if null_selection == "emit_null" else 0
    dv, dm = values.to_device(amd), mask.to_device(amd)
    out = amd.compute.filter(dv, dm, null_selection)
    want = O.binary_filter(values.offsets, values.data, values.valid_bitmap(), values.offset,
                           mask.data_bytes(), mask.valid_bitmap(), mask.offset, mask.length, code)
    tag = f"binary_filter[n={mask.length},{null_selection},voff={values.offset},moff={mask.offset}]"
    assert out.length == len(want[0]) - 1, tag
    ref = None
    if use_pyarrow and pc is not None:
        ref = pc.filter(values.to_pyarrow(), mask.to_pyarrow(), null_selection_behavior=null_selection)
    _check_binary_result(out, want, tag, ref)
    return out


def check_boolean_take_and_filter(amd, values: HostArray, indices: HostArray, mask: HostArray, use_pyarrow=True):
    """take / filter on bit-packed boolean VALUES: bits and validity vs numpy and pyarrow."""
    dv = values.to_device(amd)
    out = amd.compute.take(dv, indices.to_device(amd))
    idx = indices.logical_values().astype(np.int64)
    iv = indices.logical_valid()
    src_valid = values.logical_valid()
    want_valid = iv & src_valid[np.where(iv, idx, 0)]
    want_bits = np.where(want_valid, values.logical_values()[np.where(iv, idx, 0)], False)
    bits, pad_ok = device_bitmap_to_bool(out.data, out.length)
    assert_equal(bits, want_bits, "boolean take bits (null slots zero)")
    assert pad_ok
    gv, pad2 = _logical_valid(out)
    assert_equal(gv, want_valid, "boolean take validity")
    assert pad2 and out.null_count == int((~want_valid).sum())
    if use_pyarrow and pc is not None:
        assert out.to_pyarrow().equals(pc.take(values.to_pyarrow(), indices.to_pyarrow()))
    for sel in ("drop", "emit_null"):
        f = amd.compute.filter(dv, mask.to_device(amd), sel)
        if use_pyarrow and pc is not None:
            ref = pc.filter(values.to_pyarrow(), mask.to_pyarrow(), null_selection_behavior=sel)
            assert f.to_pyarrow().equals(ref) and f.null_count == ref.null_count, sel
        keep = mask.logical_values() & mask.logical_valid()
        if sel == "drop":
            assert f.length == int(keep.sum())
            fb, _ = device_bitmap_to_bool(f.data, f.length)
            fv, _ = _logical_valid(f)
            assert_equal(fv, src_valid[keep], "boolean filter validity")
            assert_equal(fb[fv], values.logical_values()[keep][fv], "boolean filter bits")


# ------------------------------------------------------------------ cast / compare / add
def _bits_equal_f32(got, want):
    g, w = got.view(np.uint32), want.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(want)
    return (g == w) | both_nan


def check_cast_f64_f32(amd, arr: HostArray, use_pyarrow=True):
    d = arr.to_device(amd)
    out = amd.compute.cast(d, amd.array.float32)
    want = O.cast_f64_f32(arr.logical_values())
    got = _data_np(out, np.float32)
    tag = f"cast_f64_f32[n={arr.length},off={arr.offset}]"
    ok = _bits_equal_f32(got, want)
    assert ok.all(), f"{tag}: {int((~ok).sum())} mismatches, first at {int(np.nonzero(~ok)[0][0])}"
    # tolerance stated by north_star: <= 1 ULP; we require 0 ULP except NaN payloads
    got_valid, pad_ok = _logical_valid(out)
    assert_equal(got_valid, arr.logical_valid(), tag + " validity")
    # offset-0 inputs share the bitmap zero-copy (like the reference executor), so bits past
    # `length` are the caller's; freshly written bitmaps must be zero padded
    assert pad_ok or out.validity is d.validity
    if use_pyarrow and pc is not None:
        ref = pc.cast(arr.to_pyarrow(), pa.float32())
        rv = ref.fill_null(0).to_numpy(zero_copy_only=False)
        v = arr.logical_valid()
        ok = _bits_equal_f32(got[v], rv[v])
        assert ok.all(), tag + " vs pyarrow"
    return out


def check_greater_f64(amd, left, right, use_pyarrow=True):
    """left/right: HostArray or python float (scalar broadcast)."""
    dl = left.to_device(amd) if isinstance(left, HostArray) else left
    dr = right.to_device(amd) if isinstance(right, HostArray) else right
    out = amd.compute.greater(dl, dr)
    lv = left.logical_values() if isinstance(left, HostArray) else float(left)
    rv = right.logical_values() if isinstance(right, HostArray) else float(right)
    n = out.length
    if lv is not None and np.isscalar(lv) and np.isscalar(rv):
        raise AssertionError("at least one side must be an array")
    i64 = isinstance(left, HostArray) and left.dtype == np.int64
    want_bm = O.greater_i64(lv, rv) if i64 else O.greater_f64(lv, rv)
    got_bits, pad_ok = device_bitmap_to_bool(out.data, n)
    tag = f"greater[n={n}]"
    assert_equal(got_bits, oracle_bitmap_to_bool(want_bm, n), tag + " bits")
    assert pad_ok, tag + ": padding bits not zero (scalar_compare.cc:175-188 + zeroed bitmap)"
    want_valid = np.ones(n, dtype=bool)
    for side in (left, right):
        if isinstance(side, HostArray):
            want_valid &= side.logical_valid()
    got_valid, pad2 = _logical_valid(out)
    assert_equal(got_valid, want_valid, tag + " validity")
    assert pad2
    if use_pyarrow and pc is not None:
        pl = left.to_pyarrow() if isinstance(left, HostArray) else pa.scalar(float(left), pa.float64())
        pr = right.to_pyarrow() if isinstance(right, HostArray) else pa.scalar(float(right), pa.float64())
        ref = pc.greater(pl, pr)
        rvalid = ~np.asarray(ref.is_null())
        assert_equal(got_valid, rvalid, tag + " validity vs pyarrow")
        rbits = ref.fill_null(False).to_numpy(zero_copy_only=False)
        assert_equal(got_bits[got_valid], rbits[rvalid], tag + " bits vs pyarrow")
    return out


def check_add(amd, left: HostArray, right: HostArray, use_pyarrow=True):
    out = amd.compute.add(left.to_device(amd), right.to_device(amd))
    want = O.add(left.logical_values(), right.logical_values())
    got = _data_np(out, left.dtype)
    tag = f"add[{left.dtype},n={left.length}]"
    if left.dtype.kind == "f":
        assert_equal(got.view(np.uint64), want.view(np.uint64), tag)
    else:
        assert_equal(got, want, tag)
    got_valid, _ = _logical_valid(out)
    assert_equal(got_valid, left.logical_valid() & right.logical_valid(), tag + " validity")
    if use_pyarrow and pc is not None and left.dtype.kind == "f":
        ref = pc.add(left.to_pyarrow(), right.to_pyarrow())
        rv = ref.fill_null(0).to_numpy(zero_copy_only=False)
        assert_equal(got[got_valid].view(np.uint64), rv[got_valid].view(np.uint64), tag + " vs pyarrow")
    return out


def check_scalar_operand_ops(amd, rng, n=10_000):
    """add / greater with one valid scalar operand (ScalarBinary::ArrayScalar / ScalarArray,
    codegen_internal.h; ComparePrimitiveArrayScalar, scalar_compare.cc:189-247): equal to the
    array-array oracle with the scalar broadcast, and to pyarrow."""
    for dtype, scalars in ((np.int64, [0, 7, -3, 2**62, -2**63]), (np.float64, [0.0, -1.5, 1e308, float("inf")])):
        arr = util.random_array(rng, dtype, n, null_p=0.1, offset=3, tail=2)
        if dtype == np.int64:
            arr.values[:4] = [2**63 - 1, -2**63, 2**62, -1]          # wrap-around at the edges
        d = arr.to_device(amd)
        vals, valid = arr.logical_values(), arr.logical_valid()
        for sc in scalars:
            full = np.full(n, sc, dtype=dtype)
            for out in (amd.compute.add(d, sc), amd.compute.add(sc, d)):
                got = _data_np(out, dtype)
                assert_equal(got.view(np.uint64), O.add(vals, full).view(np.uint64), f"add[{dtype.__name__}, scalar {sc}]")
                gv, pad = _logical_valid(out)
                assert_equal(gv, valid, "add scalar validity")
            cases = ((amd.compute.greater(d, sc), (vals, full)), (amd.compute.greater(sc, d), (full, vals)))
            for out, (l, r) in cases:
                want = O.greater_i64(l, r) if dtype == np.int64 else O.greater_f64(l, r)
                bits, pad_ok = device_bitmap_to_bool(out.data, n)
                assert_equal(bits, oracle_bitmap_to_bool(want, n), f"greater[{dtype.__name__}, scalar {sc}]")
                assert pad_ok
            if pc is not None and not (dtype == np.float64 and sc == 1e308):
                pa_arr = arr.to_pyarrow()
                psc = pa.scalar(sc, pa.int64() if dtype == np.int64 else pa.float64())
                ref = pc.add(pa_arr, psc)
                got = _data_np(amd.compute.add(d, sc), dtype)
                rv = ref.fill_null(0).to_numpy(zero_copy_only=False)
                assert_equal(got[valid].view(np.uint64), rv[valid].view(np.uint64), "add scalar vs pyarrow")
                refg = pc.greater(psc, pa_arr).fill_null(False).to_numpy(zero_copy_only=False)
                bits, _ = device_bitmap_to_bool(amd.compute.greater(sc, d).data, n)
                assert_equal(bits[valid], refg[valid], "greater scalar vs pyarrow")


def check_integer_casts(amd, rng, n=9000, use_pyarrow=True):
    """int64 -> int32 (checked / unsafe) and int32 -> int64."""
    A = amd.array
    a = util.random_array(rng, np.int64, n, null_p=0.1, offset=3, tail=2, lo=-2**31, hi=2**31 - 1)
    a.values[a.offset: a.offset + 2] = [2**31 - 1, -2**31]
    d = a.to_device(amd)
    want, err = O.cast_i64_i32(a.logical_values(), a.logical_valid())
    assert err is None
    for safe in (True, False):
        out = amd.compute.cast(d, A.int32, safe=safe)
        assert_equal(_data_np(out, np.int32), want, f"cast i64->i32 safe={safe}")
        gv, _ = _logical_valid(out)
        assert_equal(gv, a.logical_valid(), "cast validity")
    back = amd.compute.cast(amd.compute.cast(d, A.int32), A.int64)
    assert_equal(_data_np(back, np.int64)[a.logical_valid()], a.logical_values()[a.logical_valid()], "i32->i64 round trip")
    # out-of-range values: an error naming the FIRST offending valid slot; a null slot does not count
    bad = HostArray(a.values.copy(), None if a.valid is None else a.valid.copy(), a.offset, a.length)
    lv = bad.logical_valid()
    first_valid = int(np.nonzero(lv)[0][10])
    later = int(np.nonzero(lv)[0][500])
    nulls = np.nonzero(~lv)[0]
    bad.values[bad.offset + later] = -2**40
    bad.values[bad.offset + first_valid] = 2**31
    if len(nulls):
        bad.values[bad.offset + int(nulls[0])] = 2**50          # before both, but null
    want_bad, err = O.cast_i64_i32(bad.logical_values(), bad.logical_valid())
    assert err == "Integer value 2147483648 not in range: -2147483648 to 2147483647"
    db = bad.to_device(amd)
    with pytest.raises(amd.ArrowInvalid) as ei:
        amd.compute.cast(db, A.int32)
    assert str(ei.value) == err, str(ei.value)
    unsafe = amd.compute.cast(db, A.int32, safe=False)
    assert_equal(_data_np(unsafe, np.int32), want_bad, "unsafe cast truncates")
    if use_pyarrow and pc is not None:
        with pytest.raises(pa.lib.ArrowInvalid) as ri:
            pc.cast(bad.to_pyarrow(), pa.int32())
        assert str(ri.value) == err
        assert pc.cast(bad.to_pyarrow(), pa.int32(), safe=False).equals(unsafe.to_pyarrow())
        assert pc.cast(a.to_pyarrow(), pa.int32()).equals(amd.compute.cast(d, A.int32).to_pyarrow())


def check_cast_i64_f64(amd, rng, n=6000, use_pyarrow=True):
    A = amd.array
    a = util.random_array(rng, np.int64, n, null_p=0.1, offset=2, tail=1, lo=-2**53, hi=2**53)
    a.values[a.offset: a.offset + 2] = [2**53, -2**53]
    d = a.to_device(amd)
    want, err = O.cast_i64_f64(a.logical_values(), a.logical_valid())
    assert err is None
    out = amd.compute.cast(d, A.float64)
    assert_equal(_data_np(out, np.float64).view(np.uint64), want.view(np.uint64), "cast i64->f64")
    bad = HostArray(a.values.copy(), a.valid.copy(), a.offset, a.length)
    pos = int(np.nonzero(bad.logical_valid())[0][7])
    bad.values[bad.offset + pos] = 2**53 + 1
    want_bad, err = O.cast_i64_f64(bad.logical_values(), bad.logical_valid())
    assert err == "Integer value 9007199254740993 not in range: -9007199254740992 to 9007199254740992"
    with pytest.raises(amd.ArrowInvalid) as ei:
        amd.compute.cast(bad.to_device(amd), A.float64)
    assert str(ei.value) == err
    unsafe = amd.compute.cast(bad.to_device(amd), A.float64, safe=False)
    assert_equal(_data_np(unsafe, np.float64).view(np.uint64), want_bad.view(np.uint64), "unsafe i64->f64 rounds")
    if use_pyarrow and pc is not None:
        with pytest.raises(pa.lib.ArrowInvalid) as ri:
            pc.cast(bad.to_pyarrow(), pa.float64())
        assert str(ri.value) == err
        assert pc.cast(bad.to_pyarrow(), pa.float64(), safe=False).equals(unsafe.to_pyarrow())
        assert pc.cast(a.to_pyarrow(), pa.float64()).equals(out.to_pyarrow())


def check_scalar_aggregates(amd, rng, n=20_000):
    """sum / count / min_max of int64 columns vs numpy (exact integer arithmetic) and pyarrow."""
    for null_p, lo, hi in ((0.0, -1000, 1000), (0.2, None, None), (1.0, -5, 5)):
        a = util.random_array(rng, np.int64, n, null_p=null_p, offset=3, tail=2, lo=lo, hi=hi)
        d = a.to_device(amd)
        vals, valid = a.logical_values(), a.logical_valid()
        for skip_nulls in (True, False):
            for min_count in (0, 1, n + 1):
                exact = int(vals[valid].astype(object).sum()) if valid.any() else 0
                want_sum = ((exact + 2**63) % 2**64) - 2**63
                nulls = not valid.all()
                cnt = int(valid.sum())
                want = None if ((not skip_nulls and nulls) or cnt < min_count) else want_sum
                assert amd.compute.sum(d, skip_nulls, min_count) == want, (null_p, skip_nulls, min_count)
                want_mm = None if ((not skip_nulls and nulls) or cnt < max(1, min_count)) else (int(vals[valid].min()), int(vals[valid].max()))
                assert amd.compute.min_max(d, skip_nulls, min_count) == want_mm
                if pc is not None:
                    opts = pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count)
                    assert pc.sum(a.to_pyarrow(), options=opts).as_py() == want
                    ref = pc.min_max(a.to_pyarrow(), options=opts).as_py()
                    assert (None if ref["min"] is None else (ref["min"], ref["max"])) == want_mm
        assert amd.compute.count(d) == int(valid.sum()) and amd.compute.count(d, "only_null") == int((~valid).sum())
        assert amd.compute.count(d, "all") == n
    agg = amd.compute.Int64Aggregator(d.device)          # several batches accumulate like Consume / MergeFrom
    a = util.random_array(rng, np.int64, n, null_p=0.1)
    d = a.to_device(amd)
    for b in range(0, n, 7001):
        agg.consume(d.slice(b, 7001))
    v = a.logical_values()[a.logical_valid()]
    assert agg.sum() == ((int(v.astype(object).sum()) + 2**63) % 2**64) - 2**63 and agg.min_max() == (int(v.min()), int(v.max()))


def check_arithmetic(amd, rng, n=8000, use_pyarrow=True):
    """subtract / multiply and the *_checked forms, int64 (wrap-around, overflow only counted where
    both operands are valid) and float64; array x array and both scalar orders."""
    for dtype in (np.int64, np.float64):
        kw = {"lo": -10**6, "hi": 10**6} if dtype == np.int64 else {}
        a = util.random_array(rng, dtype, n, null_p=0.1, offset=3, tail=2, **kw)
        b = util.random_array(rng, dtype, n, null_p=0.1, offset=1, tail=4, **kw)
        da, db = a.to_device(amd), b.to_device(amd)
        la, lb = a.logical_values(), b.logical_values()
        both = a.logical_valid() & b.logical_valid()
        sc = dtype(3)
        for op in ("add", "subtract", "multiply"):
            for checked in (False, True):
                fn = getattr(amd.compute, op + ("_checked" if checked else ""))
                for out, l, r, valid in ((fn(da, db), la, lb, both), (fn(da, sc.item()), la, sc, a.logical_valid()),
                                         (fn(sc.item(), db), sc, lb, b.logical_valid())):
                    want, ovf = O.arith(op, l, r, valid)
                    assert not ovf
                    got = _data_np(out, dtype)
                    assert_equal(got.view(np.uint64)[valid], np.asarray(want).view(np.uint64)[valid], f"{op} checked={checked} {dtype.__name__}")
                    gv, _ = _logical_valid(out)
                    assert_equal(gv, valid, f"{op} validity")
                if use_pyarrow and pc is not None:
                    ref = getattr(pc, op + ("_checked" if checked else ""))(a.to_pyarrow(), b.to_pyarrow())
                    assert fn(da, db).to_pyarrow().equals(ref), (op, checked)
    # int64 edges: unchecked wraps; checked raises "overflow" — but not when the overflowing slot is null
    edge = HostArray(np.array([2**63 - 1, -2**63, 2**62, 5, -7], dtype=np.int64), None, 0, 5)
    one = HostArray(np.array([1, -1, 2, 3, 4], dtype=np.int64), None, 0, 5)
    minus = HostArray(-one.values, None, 0, 5)
    de = edge.to_device(amd)
    others = {"add": one, "subtract": minus, "multiply": one}     # each overflows in its first slots
    for op in ("add", "subtract", "multiply"):
        do = others[op].to_device(amd)
        want, ovf = O.arith(op, edge.values, others[op].values)
        assert ovf
        assert_equal(_data_np(getattr(amd.compute, op)(de, do), np.int64), want, op + " wraps")
        with pytest.raises(amd.ArrowInvalid, match="overflow"):
            getattr(amd.compute, op + "_checked")(de, do)
    masked = HostArray(edge.values.copy(), np.array([False, False, False, True, True]), 0, 5)   # the overflowing slots are null
    for op in ("add", "subtract", "multiply"):
        do = others[op].to_device(amd)
        out = getattr(amd.compute, op + "_checked")(masked.to_device(amd), do)
        want, ovf = O.arith(op, masked.values, others[op].values, masked.valid)
        assert not ovf
        assert_equal(_data_np(out, np.int64)[masked.valid], want[masked.valid], op + "_checked with null overflow slots")
    if pc is not None:
        with pytest.raises(pa.lib.ArrowInvalid, match="overflow"):
            pc.add_checked(edge.to_pyarrow(), one.to_pyarrow())
        assert pc.add_checked(masked.to_pyarrow(), one.to_pyarrow()).to_pylist()[3:] == [8, -3]


def check_numeric_compare_arith(amd, rng, dtype, n=6000, use_pyarrow=True):
    """The comparison family and add / subtract / multiply (+ _checked) on ONE numeric element type (any of int8 ...
    uint64, float32, float64): array x array and both scalar orders, sliced operands with nulls — bits / values equal
    the oracle on every valid slot and pyarrow's result as a whole; integer results wrap in the type's width; the
    checked forms raise "overflow" exactly when a slot with both operands valid overflows THAT type."""
    dt = np.dtype(dtype)
    is_int = dt.kind in "iu"
    if is_int:
        info = np.iinfo(dt)
        small = int(min(info.max, 11))          # products of two values stay inside every type
        lo = -small if dt.kind == "i" else 0
        a = util.random_array(rng, dtype, n, null_p=0.1, offset=3, tail=2, lo=lo, hi=small)
        b = util.random_array(rng, dtype, n, null_p=0.05, offset=1, tail=4, lo=lo, hi=small)
        sc = dt.type(3)
    else:
        a = util.random_array(rng, dtype, n, null_p=0.1, offset=3, tail=2)
        b = util.random_array(rng, dtype, n, null_p=0.05, offset=1, tail=4)
        a.values[:] = np.round(a.values * 2) / 2
        b.values[:] = np.round(b.values * 2) / 2
        a.values[3:9] = [np.nan, np.inf, -np.inf, 0.0, -0.0, np.nan]
        b.values[1:7] = [np.nan, np.inf, 1.0, -0.0, 0.0, 2.0]
        sc = dt.type(0.5)
    da, db = a.to_device(amd), b.to_device(amd)
    la, lb = a.logical_values(), b.logical_values()
    va, vb = a.logical_valid(), b.logical_valid()
    uview = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[dt.itemsize]
    for op in ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal"):
        fn = getattr(amd.compute, op)
        for out, want_bits, want_valid in ((fn(da, db), O.compare(op, la, lb), va & vb),
                                           (fn(da, sc.item()), O.compare(op, la, sc), va),
                                           (fn(sc.item(), db), O.compare(op, sc, lb), vb)):
            bits, pad_ok = device_bitmap_to_bool(out.data, n)
            assert pad_ok
            assert_equal(bits, want_bits, f"{op}[{dt.name}] bits")
            assert_equal(_logical_valid(out)[0], want_valid, f"{op}[{dt.name}] validity")
        if use_pyarrow and pc is not None:
            assert fn(da, db).to_pyarrow().equals(getattr(pc, op)(a.to_pyarrow(), b.to_pyarrow())), (op, dt.name)
    for op in ("add", "subtract", "multiply"):
        if is_int and dt.kind == "u" and op == "subtract":
            continue       # unsigned differences of random operands wrap / overflow: covered by the edge cases below
        for checked in (False, True):
            fn = getattr(amd.compute, op + ("_checked" if checked else ""))
            for out, l, r, valid in ((fn(da, db), la, lb, va & vb), (fn(da, sc.item()), la, sc, va),
                                     (fn(sc.item(), db), sc, lb, vb)):
                want, ovf = O.arith(op, l, r, valid, dtype=dt)
                assert not ovf
                assert out.type.name == da.type.name
                got, want = _data_np(out, dtype).copy(), np.asarray(want).copy()
                if not is_int:       # a NaN result is a NaN; its sign / payload is the FPU's (x86 and gfx950 differ)
                    got[np.isnan(got)] = np.nan
                    want[np.isnan(want)] = np.nan
                assert_equal(got.view(uview)[valid], want.view(uview)[valid], f"{op} checked={checked} {dt.name}")
                assert_equal(_logical_valid(out)[0], valid, f"{op}[{dt.name}] validity")
            if use_pyarrow and pc is not None:
                ref = getattr(pc, op + ("_checked" if checked else ""))(a.to_pyarrow(), b.to_pyarrow())
                mine = fn(da, db).to_pyarrow()
                if is_int:
                    assert mine.equals(ref), (op, checked, dt.name)
                else:      # Array.equals treats NaN != NaN: compare validity, then values with NaNs canonicalised
                    assert mine.is_valid().equals(ref.is_valid()), (op, dt.name)
                    x, y = (t.fill_null(0).to_numpy(zero_copy_only=False).copy() for t in (mine, ref))
                    x[np.isnan(x)] = np.nan
                    y[np.isnan(y)] = np.nan
                    assert_equal(x.view(uview), y.view(uview), f"{op}[{dt.name}] vs pyarrow")
    if not is_int:
        return
    # edges of the type: unchecked wraps in ITS width; checked raises "overflow" — but not under a null
    info = np.iinfo(dt)
    edge = HostArray(np.array([info.max, info.min, info.max // 2 + 1, 5, 7], dtype=dt), None, 0, 5)
    others = {"add": np.array([1, 0, info.max // 2 + 1, 3, 4], dtype=dt),
              "subtract": np.array([0, 1, 0, 3, 4], dtype=dt),
              "multiply": np.array([2, 1, 2, 3, 4], dtype=dt)}       # each overflows in slot 0, 1 or 2
    de = edge.to_device(amd)
    for op in ("add", "subtract", "multiply"):
        other = HostArray(others[op], None, 0, 5)
        do = other.to_device(amd)
        want, ovf = O.arith(op, edge.values, other.values, dtype=dt)
        assert ovf, (op, dt.name)
        assert_equal(_data_np(getattr(amd.compute, op)(de, do), dtype), want, f"{op}[{dt.name}] wraps")
        with pytest.raises(amd.ArrowInvalid, match="overflow"):
            getattr(amd.compute, op + "_checked")(de, do)
        if use_pyarrow and pc is not None:
            assert getattr(amd.compute, op)(de, do).to_pyarrow().equals(getattr(pc, op)(edge.to_pyarrow(), other.to_pyarrow()))
            with pytest.raises(pa.lib.ArrowInvalid, match="overflow"):
                getattr(pc, op + "_checked")(edge.to_pyarrow(), other.to_pyarrow())
        masked = HostArray(edge.values.copy(), np.array([False, False, False, True, True]), 0, 5)
        out = getattr(amd.compute, op + "_checked")(masked.to_device(amd), do)
        want, ovf = O.arith(op, masked.values, other.values, masked.valid, dtype=dt)
        assert not ovf
        assert_equal(_data_np(out, dtype)[masked.valid], want[masked.valid], f"{op}_checked[{dt.name}] with null overflow slots")


def check_temporal_compare(amd, rng, n=4000):
    """The comparison family on temporal columns (timestamp with / without zone, duration, time32 / time64, date32 /
    date64): equal to pyarrow's on the same arrays — bits, validity, null count; operands of different units or a zoned
    against a zone-less timestamp are refused like the reference refuses them."""
    if pa is None:
        pytest.skip("needs pyarrow")
    for t in (pa.timestamp("us"), pa.timestamp("ns", "UTC"), pa.timestamp("s", "Europe/Paris"), pa.duration("ms"),
              pa.time32("s"), pa.time64("ns"), pa.date32(), pa.date64()):
        w = 4 if t in (pa.time32("s"), pa.date32()) else 8
        raw = rng.integers(0, 80_000 if w == 4 else 10**6, n).astype(np.int32 if w == 4 else np.int64)
        if t == pa.date64():
            raw = raw * 86_400_000
        a = pa.array(raw, pa.int32() if w == 4 else pa.int64(), mask=rng.random(n) < 0.08).cast(t)
        b = pa.array(np.roll(raw, 5), pa.int32() if w == 4 else pa.int64(), mask=rng.random(n) < 0.05).cast(t)
        da, db = amd.Array.from_pyarrow(a.slice(3)), amd.Array.from_pyarrow(b.slice(3))
        for op in ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal"):
            got = amd.compute.call_function(op, [da, db])
            want = getattr(pc, op)(a.slice(3), b.slice(3))
            assert got.to_pyarrow().equals(want) and got.null_count in (want.null_count, -1), (str(t), op)
    zoned = amd.Array.from_pyarrow(pa.array([1, 2], pa.timestamp("us", "UTC")))
    naive = amd.Array.from_pyarrow(pa.array([1, 3], pa.timestamp("us")))
    with pytest.raises(amd.ArrowInvalid, match="Cannot compare timestamp with timezone to timestamp without timezone"):
        amd.compute.call_function("less", [zoned, naive])
    with pytest.raises(NotImplementedError, match="different types / units"):
        amd.compute.call_function("equal", [naive, amd.Array.from_pyarrow(pa.array([1, 3], pa.timestamp("ms")))])


def check_divide(amd, rng, n=6000, use_pyarrow=True, dtypes=(np.int64, np.float64)):
    """divide / divide_checked (Divide / DivideChecked, base_arithmetic_internal.h:366-424) on the numeric types in
    `dtypes`, array and scalar operands: results at visited slots, validity, and the error the LAST failing valid slot
    names; failing values hidden under nulls do not fail.  Signed integers: min / -1 of the type's OWN width."""
    def operands(dt):
        if dt.kind == "f":
            l = HostArray((np.round(rng.standard_normal(n + 5) * 8) / 4).astype(dt), rng.random(n + 5) >= 0.1, 3, n)
            r = HostArray(np.where(rng.random(n + 9) < 0.3, 0.0, np.round(rng.standard_normal(n + 9) * 4) / 2).astype(dt), rng.random(n + 9) >= 0.1, 7, n)
        else:
            info = np.iinfo(dt)
            l = HostArray(rng.integers(info.min // 2, info.max // 2, n + 5, dtype=dt, endpoint=True), rng.random(n + 5) >= 0.1, 3, n)
            r = HostArray(rng.integers(max(info.min, -50), min(info.max, 50), n + 9).astype(dt), rng.random(n + 9) >= 0.1, 7, n)
            if dt.kind == "i":
                l.values[3 + 11], r.values[7 + 11] = info.min, -1
        return l, r

    def run(fn_name, left, right, lhost, rhost, both, checked, dt):
        want, error = O.divide(lhost, rhost, both, checked, dtype=dt)
        fn = getattr(amd.compute, fn_name)
        if error is not None:
            with pytest.raises(amd.ArrowInvalid) as e:
                fn(left, right)
            assert str(e.value) == error, (fn_name, str(dt), str(e.value), error)
            return None
        out = fn(left, right)
        got = _data_np(out, want.dtype)
        ok = both & ~(np.isnan(want) if want.dtype.kind == "f" else np.zeros(len(want), bool))
        assert_equal(got[ok], want[ok], f"{fn_name} {dt}")
        if want.dtype.kind == "f":
            assert np.array_equal(np.isnan(got[both]), np.isnan(want[both]))
        gv, _ = _logical_valid(out)
        assert_equal(gv, both, fn_name + " validity")
        return out

    for dtype in dtypes:
        dt = np.dtype(dtype)
        is_int = dt.kind in "iu"
        l, r = operands(dt)
        both = l.logical_valid() & r.logical_valid()
        dl, dr = l.to_device(amd), r.to_device(amd)
        for checked in (False, True):
            name = "divide_checked" if checked else "divide"
            # zero divisors (and min / -1) present: the error of the last failing valid slot
            run(name, dl, dr, l.logical_values(), r.logical_values(), both, checked, dt)
            if use_pyarrow and pa is not None:
                try:
                    getattr(pc, name)(l.to_pyarrow(), r.to_pyarrow())
                    ref_error = None
                except pa.lib.ArrowInvalid as e:
                    ref_error = str(e)
                assert ref_error == O.divide(l.logical_values(), r.logical_values(), both, checked, dtype=dt)[1], (str(dt), checked, ref_error)
            # failing slots hidden under nulls: no error; every visited slot equals the oracle (and pyarrow)
            safe = HostArray(r.values.copy(), r.valid.copy(), r.offset, r.length)
            bad = (safe.values == 0) | ((safe.values == -1) if dt.kind == "i" else False)
            safe.valid[bad] = False
            both2 = l.logical_valid() & safe.logical_valid()
            out = run(name, dl, safe.to_device(amd), l.logical_values(), safe.logical_values(), both2, checked, dt)
            if use_pyarrow and pa is not None and is_int:
                assert out.to_pyarrow().equals(getattr(pc, name)(l.to_pyarrow(), safe.to_pyarrow()))
            # scalar operands on either side
            s = dt.type(7) if is_int else dt.type(0.5)
            zero = dt.type(0)
            run(name, dl, s.item(), l.logical_values(), np.asarray(s), l.logical_valid(), checked, dt)
            run(name, s.item(), safe.to_device(amd), np.asarray(s), safe.logical_values(), safe.logical_valid(), checked, dt)
            run(name, dl, zero.item(), l.logical_values(), np.asarray(zero), l.logical_valid(), checked, dt)


def check_compare_family(amd, rng, n=10_000, use_pyarrow=True):
    """equal ... less_equal on int64 and float64 (NaN, +-0, infinities), array x array and both
    scalar orders, sliced operands: bits equal the oracle on every slot, nulls propagate."""
    for dtype in (np.int64, np.float64):
        a = util.random_array(rng, dtype, n, null_p=0.1, offset=3, tail=2, **({"lo": -50, "hi": 50} if dtype == np.int64 else {}))
        b = util.random_array(rng, dtype, n, null_p=0.05, offset=1, tail=4, **({"lo": -50, "hi": 50} if dtype == np.int64 else {}))
        if dtype == np.float64:
            a.values[:] = np.round(a.values * 2) / 2
            b.values[:] = np.round(b.values * 2) / 2          # many ties
            a.values[3:9] = [np.nan, np.inf, -np.inf, 0.0, -0.0, np.nan]
            b.values[1:7] = [np.nan, np.inf, 1.0, -0.0, 0.0, 2.0]
        da, db = a.to_device(amd), b.to_device(amd)
        la, lb = a.logical_values(), b.logical_values()
        sc = dtype(0.5) if dtype == np.float64 else dtype(7)
        for op in ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal"):
            fn = getattr(amd.compute, op)
            cases = [(fn(da, db), O.compare(op, la, lb), a.logical_valid() & b.logical_valid()),
                     (fn(da, sc.item()), O.compare(op, la, sc), a.logical_valid()),
                     (fn(sc.item(), db), O.compare(op, sc, lb), b.logical_valid())]
            for out, want_bits, want_valid in cases:
                bits, pad_ok = device_bitmap_to_bool(out.data, n)
                assert_equal(bits, want_bits, f"{op}[{dtype.__name__}] bits")
                assert pad_ok
                gv, _ = _logical_valid(out)
                assert_equal(gv, want_valid, f"{op} validity")
            if use_pyarrow and pc is not None:
                ref = getattr(pc, op)(a.to_pyarrow(), b.to_pyarrow())
                assert fn(da, db).to_pyarrow().equals(ref), op


def check_kleene_and_invert(amd, left: HostArray, right: HostArray, use_pyarrow=True):
    """and_kleene / or_kleene / invert on boolean arrays: data AND validity bitmaps equal the
    oracle's word formula everywhere (also under null slots), values equal pyarrow's."""
    dl, dr = left.to_device(amd), right.to_device(amd)
    n = left.length
    lv = None if left.valid is None else left.logical_valid()
    rv = None if right.valid is None else right.logical_valid()
    for op, fn, ref_fn in (("and", amd.compute.and_kleene, "and_kleene"), ("or", amd.compute.or_kleene, "or_kleene")):
        out = fn(dl, dr)
        want_data, want_valid = O.kleene(op, left.logical_values(), lv, right.logical_values(), rv)
        bits, pad_ok = device_bitmap_to_bool(out.data, n)
        assert_equal(bits, want_data, f"{op}_kleene data")
        assert pad_ok
        gv, pad2 = _logical_valid(out)
        assert_equal(gv, want_valid, f"{op}_kleene validity")
        assert pad2
        if left.valid is None and right.valid is None:
            assert out.validity is None and out.null_count == 0
        if use_pyarrow and pc is not None:
            ref = getattr(pc, ref_fn)(left.to_pyarrow(), right.to_pyarrow())
            assert out.to_pyarrow().equals(ref), op
    inv = amd.compute.invert(dl)
    bits, pad_ok = device_bitmap_to_bool(inv.data, n)
    assert_equal(bits, ~left.logical_values(), "invert data")
    assert pad_ok
    gv, _ = _logical_valid(inv)
    assert_equal(gv, left.logical_valid(), "invert validity")
    if use_pyarrow and pc is not None:
        assert inv.to_pyarrow().equals(pc.invert(left.to_pyarrow()))


# ------------------------------------------------------------------ sort
def check_sort_indices(amd, arr: HostArray, order="ascending", null_placement="at_end",
                       use_pyarrow=True):
    d = arr.to_device(amd)
    out = amd.compute.sort_indices(d, order, null_placement)
    want = O.sort_indices(np.ascontiguousarray(arr.values), arr.valid_bitmap(), arr.offset,
                          arr.length, descending=(order == "descending"),
                          nulls_at_start=(null_placement == "at_start"))
    got = _data_np(out, np.uint64)
    tag = f"sort_indices[{arr.dtype},n={arr.length},{order},{null_placement},off={arr.offset}]"
    assert out.validity is None and out.null_count == 0
    assert_equal(got, want, tag)
    if use_pyarrow and pc is not None:
        ref = pc.array_sort_indices(arr.to_pyarrow(), order=order, null_placement=null_placement)
        assert_equal(got, ref.to_numpy(), tag + " vs pyarrow")
    return out


SORT_WIDE_RPT_DEFAULT = (24, 16)   # sort.hip g_sort_msd_wide_rpt1 / rpt2


def check_sort_wide_sampled(amd, lib, rng, n, shift, gap2=1, b2max=12, rpt=(24, 16), typed_keys=False):
    """The wide two-level sort with level-1 bucket sizes ESTIMATED from one tile in 2^shift (buckets get room to spare,
    level 2 reads what arrived) and, gap2, level-2 buckets in fixed rooms (mean + 6 sigma) instead of a histogram
    pass.  Uniform keys: the estimates must hold (strict mode turns a silent exact re-run into an error).  Sorted /
    blocky inputs where the sampled tiles say little about the rest, or whose keys are not uniform inside a level-1
    bucket: the overflow is detected and the level repeated with exact counts — same result; in strict mode the call
    must fail instead.  b2max: how many of the partition bits level 2 may take (0 = half of them, 12 = up to 4096
    second-level bins, several per thread of the scatter's counter scan).  rpt: rows per thread of the level-1 /
    level-2 scatter tiles (8: LDS-resident tile; 16 / 24: register-staged, moved through LDS in rounds)."""
    opts = {b"sort_msd": 1, b"sort_msd_segment_rows": 4096, b"sort_msd_wide": 1,
            b"sort_msd_wide_sample_shift": shift, b"sort_msd_wide_gap2": gap2, b"sort_msd_wide_b2max": b2max,
            b"sort_msd_wide_rpt1": rpt[0], b"sort_msd_wide_rpt2": rpt[1]}
    for k, v in opts.items():
        assert lib.arx_set_option(k, v) == 0
    try:
        uniform = util.random_array(rng, np.uint64, n, offset=5)
        assert lib.arx_set_option(b"sort_msd_wide_sample_strict", 1) == 0
        check_sort_indices(amd, uniform, "ascending", "at_end", use_pyarrow=False)
        signed = util.random_array(rng, np.int64, n, null_p=0.02)
        check_sort_indices(amd, signed, "descending", "at_start", use_pyarrow=False)
        # blocks of 8192 rows alternate between two narrow key ranges in a pattern the sample cannot see
        blocky = util.random_array(rng, np.uint64, n)
        v = blocky.values[blocky.offset:blocky.offset + n]
        tile = np.arange(n) // 8192
        low = (tile * 2654435761 % 7) < 3
        v[low] >>= np.uint64(3)
        ordered = util.random_array(rng, np.uint64, n)
        ordered.values[ordered.offset:ordered.offset + n] = np.sort(ordered.values[ordered.offset:ordered.offset + n])
        failed = 0
        for arr in (blocky, ordered):
            try:
                check_sort_indices(amd, arr, "ascending", "at_end", use_pyarrow=False)
            except Exception as e:   # strict: the underestimate is reported, never a wrong order
                assert "underestimated" in str(e), e
                failed += 1
        assert lib.arx_set_option(b"sort_msd_wide_sample_strict", 0) == 0
        for arr in (blocky, ordered):
            check_sort_indices(amd, arr, "ascending", "at_end", use_pyarrow=False)
        # the level-1 tiles read the caller's column themselves: every key type's loader (4- and 8-byte, float transforms)
        for dtype in (np.float32, np.int32, np.float64, np.uint32) if typed_keys else ():
            typed = util.random_array(rng, dtype, min(n, 2_000_000) // 8 + 17, null_p=0.02, offset=1)
            check_sort_indices(amd, typed, "descending" if dtype == np.int32 else "ascending", "at_end", use_pyarrow=False)
        return failed
    finally:
        lib.arx_set_option(b"sort_msd_wide_sample_strict", 0)
        lib.arx_set_option(b"sort_msd_wide_sample_shift", 4)
        lib.arx_set_option(b"sort_msd_wide_gap2", 1)
        lib.arx_set_option(b"sort_msd_wide_rpt1", SORT_WIDE_RPT_DEFAULT[0])
        lib.arx_set_option(b"sort_msd_wide_rpt2", SORT_WIDE_RPT_DEFAULT[1])
        lib.arx_set_option(b"sort_msd_wide_b2max", 11)
        lib.arx_set_option(b"sort_msd", -1)
        lib.arx_set_option(b"sort_msd_segment_rows", 1 << 27)


def check_sort_wide_many_bins(amd, lib, rng, n, bits, b2max, combos=((2, 1), (0, 1), (0, 0)), rpt=(24, 16)):
    """The wide form with the partition bits FORCED (sort_msd_wide_bits), so that few rows meet many level-2 bins: up
    to 4096 bins per level-1 partition, `per` = 2..4 counters per thread in the scatter's scan, in the bucket-start
    scan and in the fixed-room setup.  Exact and sampled sizes, fixed rooms and counted buckets, a skewed input (whole
    partitions empty, one bin holding a third of the rows), nulls, both orders."""
    opts = {b"sort_msd": 1, b"sort_msd_segment_rows": 4096, b"sort_msd_wide": 1, b"sort_msd_wide_bits": bits,
            b"sort_msd_wide_b2max": b2max, b"sort_msd_wide_rpt1": rpt[0], b"sort_msd_wide_rpt2": rpt[1]}
    for k, v in opts.items():
        assert lib.arx_set_option(k, v) == 0
    try:
        for shift, gap2 in combos:
            assert lib.arx_set_option(b"sort_msd_wide_sample_shift", shift) == 0
            assert lib.arx_set_option(b"sort_msd_wide_gap2", gap2) == 0
            uniform = util.random_array(rng, np.uint64, n, null_p=0.01, offset=3)
            check_sort_indices(amd, uniform, "ascending", "at_end", use_pyarrow=False)
            signed = util.random_array(rng, np.int64, n)
            check_sort_indices(amd, signed, "descending", "at_start", use_pyarrow=False)
            skew = util.random_array(rng, np.uint64, n)
            v = skew.values[skew.offset:skew.offset + n]
            v[: n // 3] = (v[: n // 3] & np.uint64((1 << 40) - 1)) | np.uint64(0x5A5A << 48)   # one bin, a third of the rows
            v[n // 3: n // 2] >>= np.uint64(9)                                                  # the lowest partitions only
            check_sort_indices(amd, skew, "ascending", "at_end", use_pyarrow=False)
    finally:
        lib.arx_set_option(b"sort_msd_wide_bits", 0)
        lib.arx_set_option(b"sort_msd_wide_b2max", 11)
        lib.arx_set_option(b"sort_msd_wide_rpt1", SORT_WIDE_RPT_DEFAULT[0])
        lib.arx_set_option(b"sort_msd_wide_rpt2", SORT_WIDE_RPT_DEFAULT[1])
        lib.arx_set_option(b"sort_msd_wide_sample_shift", 4)
        lib.arx_set_option(b"sort_msd_wide_gap2", 1)
        lib.arx_set_option(b"sort_msd", -1)
        lib.arx_set_option(b"sort_msd_segment_rows", 1 << 27)


def check_sort_wide_rec8(amd, lib, rng, n, bits=0, gap2=1, shift=0, rpt=(24, 16), b2max=11, wc=256, prefetch=1, l2w=1, wc_form=2,
                         wc_min_rows=1 << 17):
    """The wide form over the caller's own column moves 8-byte words {32 key bits below the level-1 digit, row id}
    (sort_msd_wide_rec8, the default): level 2 and the finish take their digits from the word, the finish ranks whole
    words, and rows whose 32 bits tie read their full keys from the column.  Cases: uniform keys (almost no tie); keys
    that agree in their top 41+ bits and differ only BELOW the word (every row of a sub-bucket ties: the order is decided by
    the column reads alone); exact duplicates (ties that stay ties: row-id order); signed descending; float64 bit patterns;
    a shared prefix that leaves no bit below the word (no column reads at all); the tie budget spent at once (the call is
    repeated with full records: same order); and rec8 switched off.  wc: level 1 write-combined by that many persistent
    workgroups (whole 128-byte lines per bin, pad words at the end of a workgroup's share; 0 = tile at a time), prefetch:
    with the next tile's keys requested early (16- or 8-row tiles) or rpt[0]-row tiles without; b2max moves partition bits
    to level 1 (more bins there); l2w: level 2 in small workgroups (1 - 3: the shapes of msdw_scatter2w_kernel, 0: the
    one-per-CU kernel of the 12-byte records reading words).  wc_form: 2 = round 6's append kernel (every store a whole
    line; wc_min_rows: rows a workgroup must have, lowered so that small inputs still run several), 1 = round 5's
    rank-and-stage kernel.  Counters say which form really ran."""
    opts = {b"sort_msd": 1, b"sort_msd_segment_rows": 4096, b"sort_msd_wide": 1, b"sort_msd_wide_bits": bits,
            b"sort_msd_wide_gap2": gap2, b"sort_msd_wide_sample_shift": shift, b"sort_msd_wide_rpt1": rpt[0],
            b"sort_msd_wide_rpt2": rpt[1], b"sort_msd_wide_rec8": 1, b"sort_msd_wide_rec8_tie_shift": 0,
            b"sort_msd_wide_b2max": b2max, b"sort_msd_wide_wc": wc, b"sort_msd_wide_wc_prefetch": prefetch,
            b"sort_msd_wide_l2w": l2w, b"sort_msd_wide_wc_form": wc_form, b"sort_msd_wide_wc_min_rows": wc_min_rows}
    for k, v in opts.items():
        assert lib.arx_set_option(k, v) == 0
    ctr = lambda name: int(lib.arx_get_counter(name))

    def run(arr, order="ascending", placement="at_end"):
        before = {c: ctr(c) for c in (b"sort_wide_runs", b"sort_wide_rec8_runs", b"sort_wide_rec8_ties", b"sort_wide_rec8_given_up",
                                      b"sort_wide_wc_runs")}
        check_sort_indices(amd, arr, order, placement, use_pyarrow=False)
        return {c.decode()[10:]: ctr(c) - v for c, v in before.items()}

    try:
        uniform = util.random_array(rng, np.uint64, n, offset=3)
        d = run(uniform)
        assert d["runs"] == 1 and d["rec8_runs"] == 1 and d["rec8_given_up"] == 0, d
        assert d["wc_runs"] == (1 if wc else 0), (d, "level 1 was expected %s" % ("write-combined" if wc else "tile at a time"))
        ordered = HostArray(np.sort(uniform.values[uniform.offset:uniform.offset + n]), None, 0, n)     # every tile feeds few bins
        d = run(ordered)
        assert d["rec8_runs"] >= 1, d
        # hb uniform top bits (about one row per value), zero bits down to bit 20, 20 random bits: rows that share their top
        # bits tie in the word and differ below it
        hb = max(8, int(np.ceil(np.log2(n))))
        hi = rng.integers(0, 1 << hb, size=n, dtype=np.uint64) << np.uint64(64 - hb)
        below = HostArray(hi | rng.integers(0, 1 << 20, size=n, dtype=np.uint64), None, 0, n)
        d = run(below)
        assert d["rec8_runs"] == 1 and d["rec8_given_up"] == 0 and d["rec8_ties"] > n // 4, d
        d = run(HostArray(below.values.view(np.int64).copy(), None, 0, n), "descending", "at_start")
        assert d["rec8_runs"] == 1 and d["rec8_given_up"] == 0 and d["rec8_ties"] > n // 4, d
        pool = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
        dups = HostArray(pool[rng.integers(0, len(pool), size=n)], None, 0, n)
        d = run(dups)
        assert d["rec8_runs"] == 1 and d["rec8_given_up"] == 0 and d["rec8_ties"] > n // 4, d
        # runs of ~60 rows that agree in the word: too many column reads per row — given up at once, full records sort them
        hi = rng.integers(0, 1 << (hb - 6), size=n, dtype=np.uint64) << np.uint64(64 - (hb - 6))
        d = run(HostArray(hi | rng.integers(0, 1 << 20, size=n, dtype=np.uint64), None, 0, n))
        assert d["runs"] == 2 and d["rec8_runs"] == 1 and d["rec8_given_up"] == 1, d
        fbits = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
        fbits[(fbits >> np.uint64(52)) & np.uint64(0x7FF) == np.uint64(0x7FF)] &= np.uint64(0xBFFFFFFFFFFFFFFF)   # (no NaN / inf: null-likes are partitioned first)
        fbits[::7] &= np.uint64(0xFFFFFFFFFFF00000)
        d = run(HostArray(fbits.view(np.float64).copy(), None, 0, n))
        assert d["rec8_runs"] >= 1, d       # (a crowded exponent may overflow a level-2 room: counted buckets, words again)
        # 32 shared leading bits + the level-1 digit + 32 bits in the word >= 64: a tie in the word is a tie of the keys
        narrow = HostArray((np.uint64(0xABCDEF12) << np.uint64(32)) | rng.integers(0, 1 << 32, size=n, dtype=np.uint64), None, 0, n)
        d = run(narrow)
        assert d["rec8_runs"] == 1 and d["rec8_ties"] == 0, d
        # the tie budget: n >> 40 = 0 rows -> the first tied row gives the attempt up, the call runs again with full records
        assert lib.arx_set_option(b"sort_msd_wide_rec8_tie_shift", 40) == 0
        d = run(below)
        assert d["runs"] == 2 and d["rec8_runs"] == 1 and d["rec8_given_up"] == 1, d
        d = run(uniform)            # (a handful of ties at most: with none the budget is never looked at)
        assert d["rec8_runs"] == 1 and d["runs"] == 1 + d["rec8_given_up"], d
        assert lib.arx_set_option(b"sort_msd_wide_rec8_tie_shift", 0) == 0
        with_nulls = util.random_array(rng, np.uint64, n, null_p=0.03)      # a row-id column beside the keys: full records
        d = run(with_nulls)
        assert d["runs"] == 1 and d["rec8_runs"] == 0, d
        assert lib.arx_set_option(b"sort_msd_wide_rec8", 0) == 0
        d = run(below)
        assert d["runs"] == 1 and d["rec8_runs"] == 0, d
    finally:
        lib.arx_set_option(b"sort_msd_wide_rec8", 1)
        lib.arx_set_option(b"sort_msd_wide_rec8_tie_shift", 4)
        lib.arx_set_option(b"sort_msd_wide_wc", 256)
        lib.arx_set_option(b"sort_msd_wide_wc_prefetch", 1)
        lib.arx_set_option(b"sort_msd_wide_wc_form", 2)
        lib.arx_set_option(b"sort_msd_wide_wc_min_rows", 1 << 17)
        lib.arx_set_option(b"sort_msd_wide_l2w", 3)
        lib.arx_set_option(b"sort_msd_wide_b2max", 11)
        lib.arx_set_option(b"sort_msd_wide_bits", 0)
        lib.arx_set_option(b"sort_msd_wide_rpt1", SORT_WIDE_RPT_DEFAULT[0])
        lib.arx_set_option(b"sort_msd_wide_rpt2", SORT_WIDE_RPT_DEFAULT[1])
        lib.arx_set_option(b"sort_msd_wide_sample_shift", 4)
        lib.arx_set_option(b"sort_msd_wide_gap2", 1)
        lib.arx_set_option(b"sort_msd", -1)
        lib.arx_set_option(b"sort_msd_segment_rows", 1 << 27)


def check_sort_limited_range(amd, lib, rng, n, wide, light=False):
    """Keys that share their top bits (row ids, timestamps, small or clustered integers): the MSD forms must take their
    digits below the shared prefix (sort_msd_prefix) — same order as the oracle with the knob on and off, ascending and
    descending, signed and unsigned, with nulls, for the hybrid form and (wide) the wide two-level form."""
    opts = {b"sort_msd": 1, b"sort_msd_segment_rows": 4096 if wide else 1 << 27, b"sort_msd_wide": 1 if wide else 0}
    for k, v in opts.items():
        assert lib.arx_set_option(k, v) == 0
    try:
        cases = [(np.uint64, 0, 1 << 40), (np.uint64, (1 << 62) + 12345, 1 << 33), (np.int64, -(1 << 35), 1 << 36),
                 (np.int64, 1_700_000_000_000_000, 86_400_000_000), (np.uint64, 0, 1 << 63), (np.int64, -5, 11),
                 (np.uint64, 77, 1)]
        for prefix in (1, 0):
            assert lib.arx_set_option(b"sort_msd_prefix", prefix) == 0
            # light (the emulator's hybrid form, where every MSD bucket is a workgroup of fibers): with the knob off only
            # the two ranges that exercise the fall-back to the LSD passes
            for dtype, lo, span in (cases if prefix or not light else cases[:1] + cases[5:6]):
                vals = (rng.integers(0, span, size=n, dtype=np.uint64).astype(np.int64) + np.int64(lo)).astype(dtype) \
                    if dtype == np.int64 else (rng.integers(0, span, size=n, dtype=np.uint64) + np.uint64(lo))
                valid = None if span == 1 else rng.random(n) >= 0.02
                arr = HostArray(vals.astype(dtype), valid, 0, n)
                for order, placement in (("ascending", "at_end"), ("descending", "at_start")):
                    check_sort_indices(amd, arr, order, placement, use_pyarrow=False)
    finally:
        lib.arx_set_option(b"sort_msd_prefix", 1)
        lib.arx_set_option(b"sort_msd", -1)
        lib.arx_set_option(b"sort_msd_segment_rows", 1 << 27)
        lib.arx_set_option(b"sort_msd_wide", 1)


# ------------------------------------------------------------------ group-by
def check_concat_arrays(amd, chunks, use_pyarrow=True):
    """Concatenate (array/concatenate.cc): chunks glued at arbitrary bit positions; values and validity equal
    the numpy concatenation of the logical rows, padding bits stay zero, pyarrow's result is the same array."""
    dev = [c.to_device(amd) for c in chunks]
    out = amd.compute.concat_arrays(dev)
    n = sum(c.length for c in chunks)
    assert out.length == n and out.offset == 0
    any_nulls = any(c.null_count() for c in chunks)
    want_valid = np.concatenate([c.logical_valid() for c in chunks]) if n else np.zeros(0, bool)
    if out.validity is not None:
        gv, pad_ok = _logical_valid(out)
        assert_equal(gv, want_valid, "concat validity")
        assert pad_ok
        # (the sum of the chunks' counts when all are known, else unknown — as Concatenate leaves it)
        assert out.null_count in (int((~want_valid).sum()), -1)
    else:
        assert not any_nulls and out.null_count == 0
    if isinstance(chunks[0], util.HostBinaryArray):
        offs, data = _binary_out(out)
        want = [x for c in chunks for x in c.logical_values()]
        got = [bytes(data[offs[i]:offs[i + 1]]) for i in range(n)]
        assert got == want, "concat binary rows"      # (null rows keep their bytes, as Concatenate copies whole ranges)
        assert offs[0] == 0
    elif chunks[0].is_bool:
        bits, pad_ok = device_bitmap_to_bool(out.data, n)
        assert_equal(bits, np.concatenate([c.logical_values() for c in chunks]) if n else np.zeros(0, bool), "concat bits")
        assert pad_ok
    else:
        assert_equal(_data_np(out, chunks[0].dtype), np.concatenate([c.logical_values() for c in chunks]), "concat data")
    if use_pyarrow and pa is not None:
        assert out.to_pyarrow().equals(pa.concat_arrays([c.to_pyarrow() for c in chunks]))


def check_order_by(amd, columns, sort_keys, null_placement="at_end", use_pyarrow=True):
    """OrderByNode::DoFinish (acero/order_by_node.cc:100-108): concatenate, SortIndices over several keys,
    Take.  `columns` = [[chunk, ...], ...] of host arrays; `sort_keys` = [(column, order)].  A row-number column
    rides along: after the sort it IS the permutation, which must equal the oracle's (and pyarrow's) indices."""
    n = sum(c.length for c in columns[0])
    row_ids = HostArray(np.arange(n, dtype=np.int64), None, 0, n)
    dev_cols = [[c.to_device(amd) for c in col] for col in columns] + [[row_ids.to_device(amd)]]
    places = [null_placement] * len(sort_keys) if isinstance(null_placement, str) else list(null_placement)
    logical = []
    for i, order in sort_keys:
        vals = np.concatenate([c.logical_values() for c in columns[i]])
        valid = np.concatenate([c.logical_valid() for c in columns[i]])
        logical.append((vals, valid))
    want = O.sort_indices_multi(logical, [o == "descending" for _, o in sort_keys], [p == "at_start" for p in places])
    got_cols = amd.compute.order_by(dev_cols, sort_keys, null_placement)
    assert_equal(_data_np(got_cols[-1], np.int64).astype(np.uint64), want, "multi-key sort permutation")
    if use_pyarrow and pa is not None:
        table = pa.table({f"c{i}": pa.chunked_array([c.to_pyarrow() for c in col]) for i, col in enumerate(columns)})
        ref_idx = pc.sort_indices(table, sort_keys=[(f"c{i}", o, p) for (i, o), p in zip(sort_keys, places)])
        assert_equal(want, ref_idx.to_numpy(), "oracle vs pyarrow sort_indices(table)")
        ref = table.take(ref_idx)
        for i, col in enumerate(got_cols[:-1]):
            g, w = col.to_pyarrow(), ref.column(i).combine_chunks()
            if pa.types.is_floating(w.type):      # (NaN != NaN for Array.equals: compare the bit patterns)
                assert np.array_equal(np.asarray(g.is_null()), np.asarray(w.is_null()))
                g, w = (pc.fill_null(x, 0.0).to_numpy(zero_copy_only=False).view(np.uint64) for x in (g, w))
                assert np.array_equal(g, w), ("order_by column", i)
            else:
                assert g.equals(w), ("order_by column", i)


def check_delta_decode(amd, rng, block_size=128, miniblocks=4):
    """arx_delta_scan_miniblocks + arx_delta_decode against the restatement of DeltaBitPackDecoder: value counts
    around block / miniblock / scan-tile boundaries, narrow, wide (wrap-around) and zero-width miniblocks."""
    vpm = block_size // miniblocks
    for n in (1, 2, vpm, vpm + 1, vpm + 2, block_size, block_size + 1, block_size + 2, 4095, 4096, 4097, 4098, 9001):
        for kind in ("walk", "full", "const", "bursty"):
            if kind == "walk":
                v = np.cumsum(rng.integers(-40, 50, n))
            elif kind == "full":
                v = rng.integers(-2**63, 2**63 - 1, n)
            elif kind == "const":
                v = np.full(n, -3)
            else:
                v = np.where(rng.random(n) < 0.01, 2**50, 1).cumsum()
            v = v.astype(np.int64)
            page = O.delta_binary_packed_encode(v, block_size, miniblocks)
            want, used = O.delta_binary_packed_decode(page + b"trailing bytes are not the decoder's")
            assert used == len(page) and np.array_equal(want, v)
            mbs, got_vpm, total, first, consumed = amd.parquet.scan_delta_miniblocks(page + b"xyz")
            assert (got_vpm, total, first, consumed) == (vpm, n, int(v[0]), len(page)), (n, kind)
            out64 = amd.parquet.decode_delta_binary_packed(page, 8)
            assert_equal(_data_np(out64, np.int64), v, f"delta int64 {kind} n={n}")
            out32 = amd.parquet.decode_delta_binary_packed(page, 4)
            assert_equal(_data_np(out32, np.int32), v.astype(np.int32), f"delta int32 {kind} n={n}")


class _Groups:
    """Canonical form of a group-by result: rows sorted by (key_is_null, key) (tests sort too,
    acero/hash_aggregate_test.cc:262-280), as numpy columns — the tuple-per-group form this replaced took most of the GPU
    tier's group-by time (a Python sort of 1.5M tuples per result).  Compares like the list of
    (key_is_null, key, sum or None) tuples it stands for; rows() builds those tuples for a message."""

    def __init__(self, null_flag, key, total, valid):
        self.null_flag, self.key, self.total, self.valid = null_flag, key, total, valid

    def __len__(self):
        return len(self.key)

    def __eq__(self, other):
        if isinstance(other, (list, tuple)):      # (the tuple form, as a golden vector spells it)
            return self.rows() == [tuple(r) for r in other]
        return (len(self) == len(other) and bool((self.null_flag == other.null_flag).all()) and bool((self.key == other.key).all())
                and bool((self.valid == other.valid).all()) and bool((self.total == other.total).all()))

    def __add__(self, other):       # (several ranks' slices, compared as one sorted result)
        if isinstance(other, list) and not other:
            return self
        return _sorted_groups(np.concatenate([self.key, other.key]), np.concatenate([1 - self.null_flag, 1 - other.null_flag]).astype(bool),
                              np.concatenate([self.total, other.total]), np.concatenate([self.valid, other.valid]))

    __radd__ = __add__

    def rows(self, lo=0, hi=None):
        hi = len(self) if hi is None else hi
        return [(int(self.null_flag[i]), int(self.key[i]), int(self.total[i]) if self.valid[i] else None) for i in range(lo, min(hi, len(self)))]

    def first_difference(self, other):
        m = min(len(self), len(other))
        bad = (self.null_flag[:m] != other.null_flag[:m]) | (self.key[:m] != other.key[:m]) | (self.valid[:m] != other.valid[:m]) | \
              (self.total[:m] != other.total[:m])
        i = int(np.argmax(bad)) if bad.any() else m
        return i, self.rows(i, i + 1), other.rows(i, i + 1)


def _sorted_groups(keys, key_valid, sums, valid):
    key_valid = np.asarray(key_valid).astype(bool)
    valid = np.asarray(valid).astype(bool)
    null_flag = (~key_valid).astype(np.uint8)
    key = np.where(key_valid, np.asarray(keys).astype(np.int64), 0)
    total = np.where(valid, np.asarray(sums).astype(np.int64), 0)
    order = np.lexsort((key, null_flag))
    return _Groups(null_flag[order], key[order], total[order], valid[order])


def check_groupby_sum(amd, keys: HostArray, values: HostArray, skip_nulls=True, min_count=1,
                      capacity=None, use_pyarrow=True, batches=1):
    opts = amd.compute.ScalarAggregateOptions(skip_nulls, min_count)
    dk, dv = keys.to_device(amd), values.to_device(amd)
    cap = capacity or max(16, 2 * keys.length + 2)
    op = amd.compute.GroupBySum(cap, dk.device, opts)
    n = keys.length
    step = max(1, (n + batches - 1) // batches)
    for b in range(0, max(n, 1), step):  # several consume calls, like ExecBatches arriving
        op.consume(dk.slice(b, min(step, n - b)), dv.slice(b, min(step, n - b)))
    gk, gkv, gs, gvalid = op.finalize()
    got = _sorted_groups(gk.cpu().numpy(), gkv.cpu().numpy(), gs.cpu().numpy(), gvalid.cpu().numpy())
    w = O.groupby_sum_i64(np.ascontiguousarray(keys.values), keys.valid_bitmap(), keys.offset,
                          np.ascontiguousarray(values.values), values.valid_bitmap(), values.offset,
                          n, skip_nulls, min_count)
    want = _sorted_groups(w["keys"], w["key_is_valid"], w["sums"], w["valid"])
    tag = f"groupby_sum[n={n},skip_nulls={skip_nulls},min_count={min_count}]"
    assert len(got) == len(want), f"{tag}: {len(got)} groups vs {len(want)}"
    assert got == want, f"{tag}: first difference (index, got, want): {got.first_difference(want)}"
    if use_pyarrow and pa is not None and n > 0:
        t = pa.table({"k": keys.to_pyarrow(), "v": values.to_pyarrow()})
        r = t.group_by("k", use_threads=False).aggregate(
            [("v", "sum", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count))])
        rk, rs = r.column("k").combine_chunks(), r.column("v_sum").combine_chunks()
        ref = _sorted_groups(rk.fill_null(0).to_numpy(zero_copy_only=False),
                             ~np.asarray(rk.is_null()),
                             rs.fill_null(0).to_numpy(zero_copy_only=False),
                             ~np.asarray(rs.is_null()))
        assert got == ref, tag + " vs pyarrow Table.group_by"
    return got


GROUP_PARTIAL = np.dtype([("sum", "<i8"), ("count", "<i8"), ("key", "<i4"), ("key_is_valid", "u1"), ("no_nulls", "u1"), ("pad", "u1", 2)])


def check_groupby_consume_partials(amd, keys: HostArray, values: HostArray, num_parts, capacity=None):
    """arx_groupby_sum_i64_consume_partials (the sharded group-by's local pass without the local table): the records of
    every region belong to that region's rank (the owner arx_groupby_export_partitioned assigns),
    the records of a key add up to the oracle's group (sum with wrap-around, count), and the owners' merges of their
    blocks give the oracle's result with every key on one rank.  Returns the number of records written."""
    import ctypes as C

    import torch

    from arrow_amd import parallel

    assert GROUP_PARTIAL.itemsize == parallel.RECORD_BYTES
    dk, dv = keys.to_device(amd), values.to_device(amd)
    n = keys.length
    cap = capacity or max(16, 2 * n + 2)
    out = parallel.consume_partials_regions(dk, dv, cap, num_parts)
    assert out is not None, "consume_partials declined"
    regions, per_part, counts = out
    counts = [int(c) for c in counts.cpu().tolist()]
    assert all(0 <= c <= per_part for c in counts), (counts, per_part)
    host = regions.cpu().numpy().view(GROUP_PARTIAL)
    blocks = [host[p * per_part: p * per_part + counts[p]] for p in range(num_parts)]
    # owners: the table path's export of the same rows
    table = amd.compute.GroupBySum(cap, dk.device)
    table.consume(dk, dv)
    recs, cnt = parallel.export_partitioned(table, num_parts)
    cnt = [int(c) for c in cnt.cpu().tolist()]
    exported = recs.cpu().numpy().view(GROUP_PARTIAL)
    owner, at = {}, 0
    for p in range(num_parts):
        for k in exported["key"][at: at + cnt[p]].tolist():
            owner[k] = p
        at += cnt[p]
    w = O.groupby_sum_i64(np.ascontiguousarray(keys.values), keys.valid_bitmap(), keys.offset,
                          np.ascontiguousarray(values.values), values.valid_bitmap(), values.offset, n, True, 0)
    kv = np.asarray(keys.values[keys.offset: keys.offset + n])
    want_sum = {int(k): int(s) for k, s in zip(w["keys"], w["sums"])}
    uniq, cnts = np.unique(kv, return_counts=True)
    want_count = dict(zip(uniq.tolist(), cnts.tolist()))
    got_sum, got_count = {}, {}
    for p, b in enumerate(blocks):
        assert np.all(b["key_is_valid"] == 1) and np.all(b["no_nulls"] == 1) and np.all(b["count"] >= 1)
        for k, s_, c_ in zip(b["key"].tolist(), b["sum"].tolist(), b["count"].tolist()):
            assert owner[k] == p, f"key {k} in region {p}, its owner is {owner[k]}"
            got_sum[k] = (got_sum.get(k, 0) + s_ + 2**63) % 2**64 - 2**63
            got_count[k] = got_count.get(k, 0) + c_
    assert got_count == want_count
    assert got_sum == want_sum
    # the receivers' side: merge_records of every block, finalize
    got = []
    for p in range(num_parts):
        owned = amd.compute.GroupBySum(max(16, 2 * counts[p] + 2), dk.device)
        block = regions[p * per_part * parallel.RECORD_BYTES: (p * per_part + counts[p]) * parallel.RECORD_BYTES]
        parallel.merge_records(owned, block)
        gk, gkv, gs, gvalid = owned.finalize()
        got += _sorted_groups(gk.cpu().numpy(), gkv.cpu().numpy(), gs.cpu().numpy(), gvalid.cpu().numpy())
    want = _sorted_groups(w["keys"], w["key_is_valid"], w["sums"], np.ones(len(w["keys"]), bool))
    assert got == want      # (_Groups.__add__ keeps the concatenation of the owners' results sorted)
    # a region too small for what arrives: ARX_CAPACITY_ERROR, not a write past the region
    lib = amd._lib.get_lib()
    stream = amd.array.current_stream(dk.device)
    ws_bytes = lib.arx_groupby_consume_workspace_bytes(n, table.capacity)
    ws = amd.array.alloc(ws_bytes + 256, dk.device)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    small = max(1, max(counts) // 2)
    guard = 64
    buf = torch.full(((num_parts * small + guard) * parallel.RECORD_BYTES,), 0xA5, dtype=torch.uint8, device=dk.device)
    cnt_dev = torch.zeros(num_parts, dtype=torch.int64, device=dk.device)
    ks, vs = dk.span(), dv.span()
    rc = lib.arx_groupby_sum_i64_consume_partials(None, table.capacity, C.byref(ks), C.byref(vs), ws_ptr,
                                                  ws.numel() - (ws_ptr - ws.data_ptr()), num_parts, buf.data_ptr(), small,
                                                  cnt_dev.data_ptr(), stream)
    if max(counts) >= 2:
        assert rc == amd._lib.ARX_CAPACITY_ERROR, rc
        assert bool((buf[num_parts * small * parallel.RECORD_BYTES:] == 0xA5).all()), "records written past the last region"
    return sum(counts)


def check_groupby_min_max(amd, keys: HostArray, values: HostArray, skip_nulls=True, capacity=None,
                          use_pyarrow=True, batches=1, with_sum=False):
    """hash_min / hash_max on the fused table vs the oracle (and pyarrow's hash_min_max)."""
    opts = amd.compute.ScalarAggregateOptions(skip_nulls, 1)
    dk, dv = keys.to_device(amd), values.to_device(amd)
    cap = capacity or max(16, 2 * keys.length + 2)
    op = amd.compute.GroupBySum(cap, dk.device, opts)
    n = keys.length
    step = max(1, (n + batches - 1) // batches)
    for b in range(0, max(n, 1), step):
        ks, vs = dk.slice(b, min(step, n - b)), dv.slice(b, min(step, n - b))
        if with_sum:
            op.consume(ks, vs)          # the sum pass over the same rows must not disturb the extrema
        op.consume_min_max(ks, vs)
    gk, gkv, gmin, gmax, gvalid = (x.cpu().numpy() for x in op.finalize_min_max())
    w = O.groupby_minmax_i64(np.ascontiguousarray(keys.values), keys.valid_bitmap(), keys.offset,
                             np.ascontiguousarray(values.values), values.valid_bitmap(), values.offset, n, skip_nulls)

    def rows(k, kv, mn, mx, valid):
        out = [(int(a) if b else None, (int(c), int(d)) if e else None)
               for a, b, c, d, e in zip(k, kv, mn, mx, valid)]
        return sorted(out, key=lambda r: (r[0] is None, r[0] or 0))

    got, want = rows(gk, gkv, gmin, gmax, gvalid), rows(w["keys"], w["key_is_valid"], w["mins"], w["maxs"], w["valid"])
    tag = f"groupby_min_max[n={n},skip_nulls={skip_nulls},batches={batches}]"
    assert len(got) == len(want), f"{tag}: {len(got)} groups vs {len(want)}"
    assert got == want, f"{tag}: first difference (index, got, want): {got.first_difference(want)}"
    if with_sum:
        _, _, sums, _ = op.finalize()   # every valid value is in exactly one group's sum
        assert int(sums.sum().item()) == int(values.logical_values()[values.logical_valid()].astype(np.int64).sum())
    if use_pyarrow and pa is not None and n > 0:
        t = pa.table({"k": keys.to_pyarrow(), "v": values.to_pyarrow()})
        r = t.group_by("k", use_threads=False).aggregate(
            [("v", "min_max", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=1))])
        rk, rmm = r.column("k").combine_chunks(), r.column("v_min_max").combine_chunks()
        ref = sorted(((a, None if b is None or b["min"] is None else (b["min"], b["max"]))
                      for a, b in zip(rk.to_pylist(), rmm.to_pylist())), key=lambda r: (r[0] is None, r[0] or 0))
        assert got == ref, tag + " vs pyarrow hash_min_max"
    return got


NUMERIC_TYPES = {"int8": np.int8, "uint8": np.uint8, "int16": np.int16, "uint16": np.uint16, "int32": np.int32,
                 "uint32": np.uint32, "int64": np.int64, "uint64": np.uint64, "float": np.float32, "double": np.float64}


def check_cast_numeric_pair(amd, rng, in_name: str, out_name: str, n: int = 3000):
    """cast(in -> out), safe and unsafe, vs the reference build (pyarrow): values on valid slots bit for bit, and
    the SAME error text ("Integer value V not in range: LO to HI" / "Float value V was truncated converting to T")
    naming the first offender.  Unsafe float -> integer of an out-of-range value is undefined behaviour in the
    reference: those slots are not compared."""
    from arrow_amd import _lib

    it, ot = NUMERIC_TYPES[in_name], NUMERIC_TYPES[out_name]
    if np.dtype(it).kind == "f":
        x = (rng.standard_normal(n) * 1e3).astype(it)
        x[::7] = np.round(x[::7])
        x[5::97] = np.nan
        x[9::131] = np.inf
    else:
        info = np.iinfo(it)
        x = rng.integers(info.min, info.max, n, dtype=it, endpoint=True)
        x[::3] = (x[::3] % 100).astype(it)
    for null_p, variant in ((0.1, "mixed"), (0.0, "small")):
        valid = rng.random(n) > null_p
        xs = x.copy()
        if variant == "small":      # values every pair can hold: the no-error path of the safe cast
            xs = (np.abs(np.nan_to_num(xs.astype(np.float64), nan=1, posinf=2, neginf=3)) % 100).astype(it)
        for safe in (True, False):
            a = amd.Array.from_numpy(xs, valid if null_p else None)
            pa_in = pa.array(xs, mask=None if not null_p else ~valid)
            try:
                want, werr = pc.cast(pa_in, pa.from_numpy_dtype(np.dtype(ot)), safe=safe), None
            except pa.lib.ArrowInvalid as e:
                want, werr = None, str(e)
            try:
                got, gerr = amd.compute.cast(a, amd.array.type_from_name(out_name), safe=safe), None
            except _lib.ArrowInvalid as e:
                got, gerr = None, str(e)
            tag = f"cast {in_name}->{out_name} safe={safe} {variant}"
            assert werr == gerr, f"{tag}: reference error {werr!r} vs {gerr!r}"
            if want is None:
                continue
            gv, gvalid = got.to_numpy()
            m = valid.copy() if null_p else np.ones(n, bool)
            if gvalid is not None:
                assert_equal(gvalid, m, tag + " validity")
            if np.dtype(it).kind == "f" and np.dtype(ot).kind in "iu" and not safe:
                info = np.iinfo(ot)
                with np.errstate(invalid="ignore"):
                    m &= np.isfinite(xs) & (xs > info.min) & (xs < info.max)
            w = np.asarray(want.fill_null(0)).astype(ot)
            assert np.array_equal(gv[m].view(np.uint8), w[m].view(np.uint8)), tag + " values"


def torch_dtype_as_signed(dt):
    """torch's unsigned 16/32/64-bit dtypes have no numpy bridge: read them through the signed type of the same width."""
    import torch

    return {torch.uint16: torch.int16, torch.uint32: torch.int32, torch.uint64: torch.int64}.get(dt, dt)


def check_groupby_sum_typed(amd, rng, key_dtype, value_dtype, n=5000):
    """hash_sum over the other integer key / value types the reference registers (Grouper key types,
    row/grouper.cc:559-611; value types + accumulators, hash_aggregate_numeric.cc:1188-1200,
    aggregate_internal.h:41-44) vs pyarrow's group_by: same groups, same sums, same output type."""
    # keys over the key type's own range where the 32-bit table can hold them (uint32 above 2^31, negative int64 that
    # fits 32 bits): the groups must come back in the CALLER's type, not as raw int32
    kinfo = np.iinfo(key_dtype)
    lo, hi = max(int(kinfo.min), -2**31), min(int(kinfo.max), 2**31 - 1 if np.dtype(key_dtype).itemsize == 8 else int(kinfo.max))
    pool = rng.integers(lo, hi, 100, dtype=np.int64, endpoint=True)
    pool[:2] = [lo, hi]
    k = pool[rng.integers(0, 100, n)].astype(key_dtype)
    info = np.iinfo(value_dtype)
    v = rng.integers(info.min, info.max, n, dtype=value_dtype, endpoint=True)
    kval, vval = rng.random(n) > 0.05, rng.random(n) > 0.1
    dk, dv = amd.Array.from_numpy(k, kval), amd.Array.from_numpy(v, vval)
    op = amd.compute.GroupBySum(1024, dk.device)
    op.consume(dk, dv)
    fk, gkv, gs, gvalid = op.finalize()
    assert str(fk.dtype).replace("torch.", "") == np.dtype(key_dtype).name, (fk.dtype, key_dtype)
    gk = fk.cpu().view(torch_dtype_as_signed(fk.dtype)).numpy().view(key_dtype) if "uint" in str(fk.dtype) and np.dtype(key_dtype).itemsize > 1 \
        else fk.cpu().numpy()
    gkv, gs, gvalid = (x.cpu().numpy() for x in (gkv, gs, gvalid))
    t = pa.table({"k": pa.array(k, mask=~kval), "v": pa.array(v, mask=~vval)})
    r = t.group_by("k", use_threads=False).aggregate([("v", "sum")])
    ref = {(None if a is None else int(a)): b for a, b in zip(r.column("k").to_pylist(), r.column("v_sum").to_pylist())}
    st = np.uint64 if op.sum_type.name == "uint64" else np.int64
    got = {(int(a) if b else None): (int(np.array(c).astype(np.int64).view(st)) if d else None)
           for a, b, c, d in zip(gk, gkv, gs, gvalid)}
    assert got == ref, (key_dtype, value_dtype)
    assert str(r.column("v_sum").type) == op.sum_type.name


def replay_golden_sort(amd, gold, dtype):
    """The golden sort cases of tests/golden/reference_vectors.json (vector_sort_test.cc:640-724) on the device."""
    ran = 0
    wide = gold["sort_indices_narrow_and_wide"]["int64"] if np.dtype(dtype) == np.dtype(np.int64) else []   # (SortInt64, :867-885)
    for case in gold["sort_indices_integral"] + gold["sort_indices_real"] + wide:
        vals = case["values"]
        kind = np.dtype(dtype).kind
        if kind != "f" and any(x == "NaN" or (isinstance(x, float) and x != int(x)) for x in vals if x is not None):
            continue
        valid = np.array([x is not None for x in vals], dtype=bool)
        arr = np.array([0 if x is None else (np.nan if x == "NaN" else x) for x in vals], dtype=dtype)
        if len(vals) == 0:
            d = amd.Array.from_numpy(np.zeros(0, dtype=dtype))
        else:
            d = amd.Array.from_numpy(arr, None if valid.all() else valid)
        got = amd.compute.sort_indices(d, order=case["order"], null_placement=case["null_placement"])
        assert got.to_numpy()[0].tolist() == case["want"], (case, dtype)
        ran += 1
    return ran


def replay_golden_sum_only(amd, gold):
    """SumOnly (acero/hash_aggregate_test.cc:839-883): one consume per batch, key-sorted result."""
    g = gold["hash_sum_sum_only"]
    op = amd.compute.GroupBySum(64)
    for b in g["batches"]:
        kv = np.array([x is not None for x in b["key"]])
        vv = np.array([x is not None for x in b["argument"]])
        op.consume(amd.Array.from_numpy(np.array([0 if x is None else x for x in b["key"]], dtype=np.int32), kv),
                   amd.Array.from_numpy(np.array([0 if x is None else x for x in b["argument"]], dtype=np.int64), vv))
    gk, gkv, gs, gvalid = (x.cpu().numpy() for x in op.finalize())
    rows = sorted(((int(a) if b else None, int(c) if d else None) for a, b, c, d in zip(gk, gkv, gs, gvalid)),
                  key=lambda r: (r[0] is None, r[0] or 0))
    assert [list(r) for r in rows] == g["want_sorted_by_key"]


# ------------------------------------------------------------------ Grouper
_GROUPER_NP = {"uint8": np.uint8, "int8": np.int8, "uint16": np.uint16, "int16": np.int16, "uint32": np.uint32,
               "int32": np.int32, "uint64": np.uint64, "int64": np.int64, "float32": np.float32, "float64": np.float64}


def _golden_key_columns(rows, dtypes):
    """[[k0, k1, ...] per row] with None for null -> one (values, valid) pair per key column."""
    cols = []
    for j, dt in enumerate(dtypes):
        vals = np.array([0 if r[j] is None else float(r[j]) if isinstance(r[j], str) else r[j] for r in rows], dtype=dt)
        valid = np.array([r[j] is not None for r in rows], dtype=bool)
        cols.append((vals, valid))
    return cols


def _uniques_rows(uniq_cols):
    """(values, valid) per column -> list of rows with None for null, floats as their bytes."""
    n = len(uniq_cols[0][0])
    return [tuple(None if not valid[i] else vals[i:i + 1].tobytes() for vals, valid in uniq_cols) for i in range(n)]


def replay_golden_grouper(gold, section, make_grouper, max_type_combos=None):
    """Replays the reference's own Grouper tests (ExpectConsume / ExpectPopulate / ExpectLookup / ExpectUniques,
    compute/row/grouper_test.cc) on `make_grouper(dtypes) -> (consume, lookup, uniques)`; used with the oracle and
    with the device Grouper (emulator, GPU).  Returns the number of type combinations run."""
    import itertools

    g = gold[section]
    width = len(g["sequences"][0][0]["keys"][0])
    combos = list(itertools.product(g["types"], repeat=width))
    combos = [c for c in combos if sum(np.dtype(_GROUPER_NP[t]).itemsize for t in c) <= 16]
    if max_type_combos is not None:
        step = max(1, len(combos) // max_type_combos)
        combos = combos[::step]
    for combo in combos:
        dtypes = [_GROUPER_NP[t] for t in combo]
        for seq in g["sequences"]:
            consume, lookup, uniques = make_grouper(dtypes)
            for step in seq:
                cols = _golden_key_columns(step["keys"], dtypes)
                tag = f"{section}{combo} {step['op']} {step['keys']}"
                if step["op"] == "lookup":
                    ids, found = lookup(cols)
                    got = [int(i) if f else None for i, f in zip(ids, found)]
                    assert got == step["ids"], tag
                else:
                    ids = consume(cols)
                    if step["op"] == "consume":
                        assert [int(i) for i in ids] == step["ids"], tag
                if "uniques" in step:
                    want = _uniques_rows(_golden_key_columns(step["uniques"], dtypes))
                    assert _uniques_rows(uniques(dtypes)) == want, tag + " uniques"
    return len(combos)


def oracle_grouper_factory(dtypes):
    g = O.Grouper(len(dtypes))
    return (lambda cols: g.consume(cols)), (lambda cols: g.lookup(cols)), (lambda dts: g.uniques(dts))


def device_grouper_factory(amd, max_groups=64):
    def make(dtypes):
        from arrow_amd.array import type_from_numpy

        types = [type_from_numpy(np.dtype(dt)) for dt in dtypes]
        g = amd.compute.Grouper(types, max_groups)

        def up(cols):
            return [amd.Array.from_numpy(v, valid if not valid.all() else None) for v, valid in cols]

        def consume(cols):
            return _data_np(g.consume(up(cols)), np.uint32)

        def lookup(cols):
            out = g.lookup(up(cols))
            found, pad_ok = _logical_valid(out)
            assert pad_ok
            return _data_np(out, np.uint32), found

        def uniques(dts):
            batch = g.get_uniques()
            assert batch.length == g.num_groups
            return [(_data_np(a, dt).copy(), _logical_valid(a)[0]) for a, dt in zip(batch.values, dts)]

        return consume, lookup, uniques

    return make


def check_grouper(amd, rng, dtypes, n, cardinality, null_p=0.0, batches=1, max_groups=None):
    """Random key rows through the device Grouper in `batches` consumes: the ids equal the oracle's (first appearance
    in row order, across batches), uniques[id] is the key row, num_groups matches; then a Lookup of a mix of seen and
    unseen rows."""
    from arrow_amd.array import type_from_numpy

    cols = []
    for dt in dtypes:
        info = np.iinfo(dt) if np.issubdtype(dt, np.integer) else None
        pool = (rng.integers(info.min, info.max, size=cardinality, dtype=dt, endpoint=True) if info is not None
                else rng.standard_normal(cardinality).astype(dt))
        vals = pool[rng.integers(0, cardinality, size=n)]
        valid = rng.random(n) >= null_p if null_p else np.ones(n, dtype=bool)
        cols.append((vals, valid))
    want_ids, first = O.grouper_ids_one_batch(cols)
    num_groups = len(first)
    g = amd.compute.Grouper([type_from_numpy(np.dtype(dt)) for dt in dtypes], max_groups or max(16, num_groups))
    step = max(1, (n + batches - 1) // batches)
    got = []
    for b in range(0, n, step):
        part = [amd.Array.from_numpy(v[b:b + step], None if valid[b:b + step].all() else valid[b:b + step])
                for v, valid in cols]
        got.append(_data_np(g.consume(part), np.uint32).copy())
    got = np.concatenate(got) if got else np.zeros(0, np.uint32)
    tag = f"grouper[{[np.dtype(d).name for d in dtypes]},n={n},card={cardinality},null_p={null_p},batches={batches}]"
    assert_equal(got, want_ids, tag + " ids")
    assert g.num_groups == num_groups, tag
    uniq = g.get_uniques()
    for (vals, valid), arr, dt in zip(cols, uniq.values, dtypes):
        uv, pad_ok = _logical_valid(arr)
        assert pad_ok
        assert_equal(uv, valid[first], tag + " uniques validity")
        gotv = _data_np(arr, dt)
        assert_equal(gotv[uv].view(np.uint8), vals[first][uv].view(np.uint8), tag + " uniques values")
        assert arr.null_count == int((~valid[first]).sum())
    # Lookup: every second row replaced by a fresh random row (almost surely unseen when the key space is wide)
    if n:
        probe = []
        for (vals, valid), dt in zip(cols, dtypes):
            v = vals.copy()
            info = np.iinfo(dt) if np.issubdtype(dt, np.integer) else None
            fresh = (rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True) if info is not None
                     else rng.standard_normal(n).astype(dt))
            v[::2] = fresh[::2]
            probe.append((v, valid))
        orc = O.Grouper(len(dtypes))
        # (the oracle class is row-at-a-time; seed it with the unique rows in id order instead of all n rows)
        orc.consume([(vals[first], valid[first]) for vals, valid in cols])
        want_l, want_f = orc.lookup(probe)
        out = g.lookup([amd.Array.from_numpy(v, None if valid.all() else valid) for v, valid in probe])
        found, pad_ok = _logical_valid(out)
        assert pad_ok
        assert_equal(found, want_f, tag + " lookup validity")
        assert_equal(_data_np(out, np.uint32)[found], want_l[want_f], tag + " lookup ids")
        assert g.num_groups == num_groups   # Lookup adds nothing
    return g


def check_grouper_chain(amd):
    """Key rows wider than one 16-byte table go through a chain of tables (arrow_amd.compute.Grouper): how the columns
    are split, and Lookups of rows whose PREFIX is known but whose tail is not (and the other way round) — a level's
    null id must make every later level miss."""
    from arrow_amd.array import int64, int32, uint8
    from arrow_amd.compute import _grouper_levels

    assert _grouper_levels([8]) == [[0]] and _grouper_levels([8, 8]) == [[0, 1]]
    assert _grouper_levels([8, 8, 8]) == [[0, 1], [2]]
    assert _grouper_levels([8, 8, 8, 4, 8]) == [[0, 1], [2, 3], [4]]          # 4 + 8 + 4 = 16 fits, + 8 does not
    assert _grouper_levels([1] * 20) == [list(range(8)), list(range(8, 15)), list(range(15, 20))]   # 8 columns a table
    g = amd.compute.Grouper([int64, int64, int32, uint8], 64)
    assert g.num_levels == 2

    def up(rows, valid=None):
        cols = [np.array([r[j] for r in rows], dtype=dt) for j, dt in enumerate((np.int64, np.int64, np.int32, np.uint8))]
        return [amd.Array.from_numpy(c, None if valid is None else np.array([v[j] for v in valid])) for j, c in enumerate(cols)]

    rows = [(1, 2, 3, 4), (1, 2, 3, 5), (9, 2, 3, 4), (1, 2, 3, 4), (9, 2, 3, 4), (1, 2, 7, 4)]
    assert _data_np(g.consume(up(rows)), np.uint32).tolist() == [0, 1, 2, 0, 2, 3]
    assert g.num_groups == 4
    probes = [(1, 2, 3, 4), (1, 2, 3, 6), (8, 2, 3, 4), (9, 2, 3, 5), (9, 2, 3, 4), (1, 2, 7, 4), (2, 1, 7, 4)]
    out = g.lookup(up(probes))
    found, pad_ok = _logical_valid(out)
    assert pad_ok and found.tolist() == [True, False, False, False, True, True, False]
    assert _data_np(out, np.uint32)[found].tolist() == [0, 2, 3]
    assert g.num_groups == 4
    # a null is a key value of its own at every level; the second batch continues the numbering
    more = [(1, 2, 3, 4), (1, 0, 3, 4), (1, 0, 3, 4), (0, 2, 3, 0), (1, 2, 3, 0)]
    valid = [(1, 1, 1, 1), (1, 0, 1, 1), (1, 0, 1, 1), (0, 1, 1, 0), (1, 1, 1, 0)]
    valid = [tuple(bool(x) for x in v) for v in valid]
    assert _data_np(g.consume(up(more, valid)), np.uint32).tolist() == [0, 4, 4, 5, 6]
    uniq = g.get_uniques()
    want_rows = rows[:3] + [rows[5]] + more[1:2] + more[3:]
    want_valid = [(True,) * 4] * 4 + [valid[1], valid[3], valid[4]]
    for j, (arr, dt) in enumerate(zip(uniq.values, (np.int64, np.int64, np.int32, np.uint8))):
        uv, pad_ok = _logical_valid(arr)
        assert pad_ok and uv.tolist() == [v[j] for v in want_valid], j
        assert _data_np(arr, dt)[uv].tolist() == [r[j] for r, v in zip(want_rows, want_valid) if v[j]], j
    g.reset()
    assert g.num_groups == 0 and _data_np(g.consume(up(rows[2:4])), np.uint32).tolist() == [0, 1]


def check_binary_key_columns(amd, values, rng):
    """arx_binary_key_lengths / _chunk / arx_group_first_rows against their definition in numpy, then the use they are
    made for: Grouper over (length, chunks...) of a utf8 / binary column groups exactly the equal strings (null a key of
    its own, "" another), ids in order of first appearance, take(values, first_rows) = the unique strings."""
    from arrow_amd.array import uint32, uint64

    dv = values.to_device(amd)
    lens, chunks = amd.compute.binary_key_columns(dv)
    n, off = values.length, values.offset
    o = values.offsets.astype(np.int64)
    ok = values.logical_valid()
    want_len = np.where(ok, o[off + 1:off + n + 1] - o[off:off + n], 0xFFFFFFFF).astype(np.uint32)
    assert_equal(_data_np(lens, np.uint32), want_len, "binary key lengths")
    max_len = int(want_len[ok].max()) if ok.any() else 0
    assert len(chunks) == (max_len + 11) // 12
    rows = [bytes(values.data[o[off + i]:o[off + i + 1]]) if ok[i] else None for i in range(n)]
    for c, (lo, hi) in enumerate(chunks):
        want = np.zeros((n, 12), dtype=np.uint8)
        for i, r in enumerate(rows):
            part = (r or b"")[12 * c:12 * c + 12]
            want[i, :len(part)] = np.frombuffer(part, dtype=np.uint8)
        assert_equal(_data_np(lo, np.uint64), want[:, :8].copy().view("<u8").ravel(), f"binary key chunk {c} lo")
        assert_equal(_data_np(hi, np.uint32), want[:, 8:].copy().view("<u4").ravel(), f"binary key chunk {c} hi")
    # the Grouper over the virtual columns (the mirror's Grouper chains up to 32 columns: strings of <= 180 bytes; longer
    # ones only go through the one-pass route below, over ids made here)
    cols = [lens] + [h for pair in chunks for h in pair]
    first_of = {}
    want_ids = np.array([first_of.setdefault(r, len(first_of)) for r in rows], dtype=np.uint32)
    if len(cols) <= 32:
        g = amd.compute.Grouper([c.type for c in cols], max(16, n))
        ids = g.consume(cols)
        assert_equal(_data_np(ids, np.uint32), want_ids, "ids over the virtual key columns")
        assert g.num_groups == len(first_of)
    else:
        ids = amd.Array.from_numpy(want_ids)
    first = amd.compute.group_first_rows(ids, len(first_of))
    want_first = np.array([rows.index(r) for r in first_of], dtype=np.uint32)
    assert_equal(_data_np(first, np.uint32), want_first, "group first rows")
    if n:
        uniq = amd.compute.take(dv, first, boundscheck=False)
        uo = uniq.buffers[1].cpu().numpy().view(np.uint8)[:(uniq.length + 1) * 4].view(np.int32)
        ud = uniq.buffers[2].cpu().numpy().view(np.uint8) if uniq.buffers[2] is not None else np.zeros(0, np.uint8)
        uv, _ = _logical_valid(uniq)
        got = [bytes(ud[uo[i]:uo[i + 1]]) if uv[i] else None for i in range(uniq.length)]
        assert got == list(first_of), "unique strings"
    # ---- round 4: one pass whatever the lengths — (length, 64-bit hash) + verification against the first rows' bytes
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    sp = dv.binary_span() if hasattr(dv, "binary_span") else None
    if sp is None:
        sp = _lib.ArxBinarySpan(dv.buffers[0].data_ptr() if dv.buffers[0] is not None and dv.null_count != 0 else None, dv.buffers[1].data_ptr(),
                                dv.buffers[2].data_ptr() if dv.buffers[2] is not None else None, dv.offset, n, dv.null_count if dv.null_count is not None else -1)
    hashes = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    _lib.check(lib.arx_binary_key_hash(C.byref(sp), 64, hashes.data_ptr(), st))
    h = hashes.cpu().numpy()[:n].view(np.uint64)
    by_row = {}
    for i, r in enumerate(rows):
        if r is None:
            assert h[i] == 0, "a null hashes to 0"
        else:
            assert by_row.setdefault(r, h[i]) == h[i], "equal strings, different hashes"
    assert len(set(by_row.values())) == len(by_row), "64-bit hashes of a few thousand distinct strings collide"
    if n:
        ws = torch.zeros(8, dtype=torch.int64, device=dev)
        bad = C.c_int64(-1)
        _lib.check(lib.arx_binary_key_verify(C.byref(sp), ids.data.data_ptr(), first.data.data_ptr(), C.byref(bad), ws.data_ptr(), st))
        assert bad.value == 0, ("exact groups fail the verification", bad.value)
        # every row in ONE group: exactly the rows that differ from row 0 are reported
        zeros = torch.zeros(n, dtype=torch.int32, device=dev)
        _lib.check(lib.arx_binary_key_verify(C.byref(sp), zeros.data_ptr(), zeros.data_ptr(), C.byref(bad), ws.data_ptr(), st))
        assert bad.value == sum(1 for r in rows if r != rows[0]), ("verification count", bad.value)
        # a 2-bit hash: still equal for equal strings
        _lib.check(lib.arx_binary_key_hash(C.byref(sp), 2, hashes.data_ptr(), st))
        assert int(hashes.cpu().numpy()[:n].view(np.uint64).max()) < 4


def check_group_by_keys(amd, rng, key_dtypes, n, cardinality, null_p=0.0, use_pyarrow=True):
    """compute.group_by over several / wide key columns (Grouper + the dense hash_sum state): sum, count and mean per
    group equal the oracle's per-group reduction and pyarrow's Table.group_by(keys).aggregate (compared as a mapping
    from key row to results: the reference's fast grouper does not promise an output order)."""
    cols = []
    for dt in key_dtypes:
        info = np.iinfo(dt)
        pool = rng.integers(info.min, info.max, size=cardinality, dtype=dt, endpoint=True)
        vals = pool[rng.integers(0, cardinality, size=n)]
        valid = rng.random(n) >= null_p if null_p else np.ones(n, dtype=bool)
        cols.append((vals, valid))
    values = rng.integers(-10**9, 10**9, size=n, dtype=np.int64)
    vvalid = rng.random(n) >= 0.1
    keys = [amd.Array.from_numpy(v, None if valid.all() else valid) for v, valid in cols]
    varr = amd.Array.from_numpy(values, vvalid)
    uniq, (sums, counts, means) = amd.compute.group_by(keys, [(varr, "hash_sum"), (varr, "hash_count"), (varr, "hash_mean")])
    ids, first = O.grouper_ids_one_batch(cols)
    g = len(first)
    want_sum = np.zeros(g, dtype=np.int64)
    np.add.at(want_sum, ids[vvalid], values[vvalid])
    want_cnt = np.bincount(ids[vvalid], minlength=g).astype(np.int64)
    tag = f"group_by[{[np.dtype(d).name for d in key_dtypes]},n={n},card={cardinality}]"
    assert uniq.length == g and sums.length == g
    for (vals, valid), arr, dt in zip(cols, uniq.values, key_dtypes):
        uv, _ = _logical_valid(arr)
        assert_equal(uv, valid[first], tag + " key validity")
        assert_equal(_data_np(arr, dt)[uv], vals[first][uv], tag + " keys")
    sv, _ = _logical_valid(sums)
    assert_equal(sv, want_cnt >= 1, tag + " sum validity")      # min_count = 1: groups without a valid value are null
    assert_equal(_data_np(sums, np.int64)[sv], want_sum[sv], tag + " sums")
    assert_equal(_data_np(counts, np.int64), want_cnt, tag + " counts")
    mv, _ = _logical_valid(means)
    assert_equal(mv, want_cnt >= 1, tag + " mean validity")
    with np.errstate(invalid="ignore", divide="ignore"):
        want_mean = want_sum.astype(np.float64) / want_cnt.astype(np.float64)
    assert_equal(_data_np(means, np.float64)[mv].view(np.uint64), want_mean[mv].view(np.uint64), tag + " means")
    if use_pyarrow and pa is not None:
        names = [f"k{j}" for j in range(len(cols))]
        t = pa.table({**{nm: pa.array(v, mask=~valid) for nm, (v, valid) in zip(names, cols)},
                      "v": pa.array(values, mask=~vvalid)})
        ref = t.group_by(names, use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "mean")]).to_pydict()
        ref_map = {tuple(ref[nm][i] for nm in names): (ref["v_sum"][i], ref["v_count"][i], ref["v_mean"][i])
                   for i in range(len(ref["v_sum"]))}
        got_keys = []
        for a, dt in zip(uniq.values, key_dtypes):      # (ONE device read-back per column: it used to be one per row)
            col = _data_np(a, dt)[:g].astype(object)
            col[~_logical_valid(a)[0][:g]] = None
            got_keys.append(col.tolist())
        gs, gc, gm = _data_np(sums, np.int64), _data_np(counts, np.int64), _data_np(means, np.float64)
        got_map = {tuple(col[i] for col in got_keys): (int(gs[i]) if sv[i] else None, int(gc[i]),
                                                       float(gm[i]) if mv[i] else None) for i in range(g)}
        assert got_map == ref_map, tag + " vs pyarrow"


def check_groupby_mean(amd, keys: HostArray, values: HostArray, skip_nulls=True, min_count=1, capacity=None,
                       use_pyarrow=True, batches=1, expect_decline=False):
    """hash_mean(int64) on the fused table vs the oracle's row-order double accumulation (and pyarrow's hash_mean):
    bit-exact doubles wherever every partial sum is an exact integer; declines (NotImplemented) otherwise."""
    opts = amd.compute.ScalarAggregateOptions(skip_nulls, min_count)
    dk, dv = keys.to_device(amd), values.to_device(amd)
    cap = capacity or max(16, 2 * keys.length + 2)
    op = amd.compute.GroupBySum(cap, dk.device, opts)
    n = keys.length
    step = max(1, (n + batches - 1) // batches)
    for b in range(0, max(n, 1), step):
        ks, vs = dk.slice(b, min(step, n - b)), dv.slice(b, min(step, n - b))
        op.consume(ks, vs)
        op.consume_min_max(ks, vs)
    if expect_decline:
        import pytest

        with pytest.raises(NotImplementedError, match="2\\^53"):
            op.finalize_mean()
        return None
    gk, gkv, gmean, gvalid = (x.cpu().numpy() for x in op.finalize_mean())
    w = O.groupby_mean_i64(np.ascontiguousarray(keys.values), keys.valid_bitmap(), keys.offset,
                           np.ascontiguousarray(values.values), values.valid_bitmap(), values.offset, n, skip_nulls,
                           min_count)

    def rows(k, kv, m, valid):
        out = [(int(a) if b else None, np.float64(c).tobytes() if e else None) for a, b, c, e in zip(k, kv, m, valid)]
        return sorted(out, key=lambda r: (r[0] is None, r[0] or 0))

    got, want = rows(gk, gkv, gmean, gvalid), rows(w["keys"], w["key_is_valid"], w["means"], w["valid"])
    tag = f"groupby_mean[n={n},skip_nulls={skip_nulls},min_count={min_count},batches={batches}]"
    assert len(got) == len(want), f"{tag}: {len(got)} groups vs {len(want)}"
    for i, (g, x) in enumerate(zip(got, want)):
        assert g == x, f"{tag}: group {i}: got {g} want {x} (bit patterns of the float64 means)"
    if use_pyarrow and pa is not None and n > 0:
        t = pa.table({"k": keys.to_pyarrow(), "v": values.to_pyarrow()})
        r = t.group_by("k", use_threads=False).aggregate(
            [("v", "mean", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count))])
        rk, rm = r.column("k").combine_chunks(), r.column("v_mean").combine_chunks()
        ref = sorted(((a, None if b is None else np.float64(b).tobytes()) for a, b in zip(rk.to_pylist(), rm.to_pylist())),
                     key=lambda r: (r[0] is None, r[0] or 0))
        assert got == ref, tag + " vs pyarrow hash_mean"
    return got


def check_unique_and_value_counts(amd, arr: HostArray, use_pyarrow=True):
    """unique / value_counts (first-appearance order) vs the oracle and pyarrow."""
    d = arr.to_device(amd)
    want_v, want_ok, want_c = O.unique_i32(arr.values, arr.valid_bitmap(), arr.offset, arr.length, True)
    u2, c = amd.compute.value_counts(d)
    for u in (amd.compute.unique(d), u2):
        vals, valid = u.to_numpy()
        valid = np.ones(u.length, bool) if valid is None else valid
        assert u.length == len(want_v), (u.length, len(want_v))
        assert_equal(valid, want_ok, "unique validity")
        assert_equal(vals[valid], want_v[want_ok], "unique values (first-appearance order)")
        assert u.null_count == int((~want_ok).sum())
    assert_equal(c.to_numpy()[0], want_c, "value_counts counts")
    if use_pyarrow and pc is not None:
        rc = pc.value_counts(arr.to_pyarrow())
        assert u2.to_pyarrow().equals(rc.field("values")) and c.to_pyarrow().equals(rc.field("counts"))
        assert u2.to_pyarrow().equals(pc.unique(arr.to_pyarrow()))


def check_dictionary_encode(amd, arr: HostArray, use_pyarrow=True):
    d = arr.to_device(amd)
    for mode in ("mask", "encode"):
        idx, dic = amd.compute.dictionary_encode(d, null_encoding=mode)
        wi, wiv, wd, wdv = O.dictionary_encode_i32(arr.values, arr.valid_bitmap(), arr.offset, arr.length, mode == "encode")
        gi, giv = idx.to_numpy()
        giv = np.ones(idx.length, bool) if giv is None else giv
        assert_equal(giv, wiv, f"dictionary_encode[{mode}] index validity")
        assert_equal(gi[giv], wi[wiv], f"dictionary_encode[{mode}] indices")
        gd, gdv = dic.to_numpy()
        gdv = np.ones(dic.length, bool) if gdv is None else gdv
        assert dic.length == len(wd)
        assert_equal(gdv, wdv, "dictionary validity")
        assert_equal(gd[gdv], wd[wdv], "dictionary values (first-appearance order)")
        if use_pyarrow and pc is not None and arr.length:
            ref = pc.dictionary_encode(arr.to_pyarrow(), null_encoding=mode)
            assert idx.to_pyarrow().equals(ref.indices) and dic.to_pyarrow().equals(ref.dictionary), mode


# ------------------------------------------------------------------ hash_sum kernel vtable
def check_hash_sum_kernel(amd, rng, n=5000, num_groups=37, null_p=0.2, skip_nulls=True, min_count=1,
                          use_pyarrow=True):
    """Drives hash_sum's HashAggregateKernel {init, resize, consume, merge, finalize} the way
    GroupByNode does (acero/groupby_aggregate_node.cc:210-337): two thread-local states fed
    batches with dense uint32 group ids, merged through a group_id_mapping, finalized.
    Compared with the oracle's restatement and with the reference (pyarrow Table.group_by)."""
    k = amd.compute.get_function_registry().get_function("hash_sum").dispatch_exact(
        [amd.array.int64, amd.array.uint32])
    opts = amd.compute.ScalarAggregateOptions(skip_nulls, min_count)
    vals = util.random_array(rng, np.int64, n, null_p=null_p, offset=3)
    gid_a = rng.integers(0, num_groups, size=n).astype(np.uint32)
    # state B sees only a subset of the groups, numbered differently (like another thread's grouper)
    nb = max(1, num_groups // 2)
    perm = rng.permutation(num_groups)[:nb].astype(np.uint32)     # B's group j == A's group perm[j]
    gid_b_local = rng.integers(0, nb, size=n).astype(np.uint32)
    vals_b = util.random_array(rng, np.int64, n, null_p=null_p, offset=0)

    dev_vals, dev_vals_b = vals.to_device(amd), vals_b.to_device(amd)
    ga = HostArray(gid_a, None, 0, n).to_device(amd)
    gb = HostArray(gid_b_local, None, 0, n).to_device(amd)
    sa, sb = k.init(opts, dev_vals.device), k.init(opts, dev_vals.device)
    half = n // 2
    k.resize(sa, num_groups // 2 + 1)          # groups appear over time: resize grows the state
    first = gid_a[:half] < (num_groups // 2 + 1)
    # batch 1 only touches already-resized groups
    sel = np.nonzero(first)[0]
    if len(sel):
        b1v = HostArray(vals.values[vals.offset:vals.offset + half][first].copy(),
                        None if vals.valid is None else vals.valid[vals.offset:vals.offset + half][first].copy(),
                        0, len(sel)).to_device(amd)
        b1g = HostArray(gid_a[:half][first].copy(), None, 0, len(sel)).to_device(amd)
        k.consume(sa, [b1v, b1g])
    k.resize(sa, num_groups)
    rest = np.nonzero(~first)[0]
    if len(rest):
        b2v = HostArray(vals.values[vals.offset:vals.offset + half][~first].copy(),
                        None if vals.valid is None else vals.valid[vals.offset:vals.offset + half][~first].copy(),
                        0, len(rest)).to_device(amd)
        b2g = HostArray(gid_a[:half][~first].copy(), None, 0, len(rest)).to_device(amd)
        k.consume(sa, [b2v, b2g])
    k.consume(sa, [dev_vals.slice(half, n - half), ga.slice(half, n - half)])   # sliced: offsets != 0
    k.resize(sb, nb)
    k.consume(sb, [dev_vals_b, gb])
    k.consume(sb, [amd.array.Scalar(7, amd.array.int64), gb.slice(0, 100)])     # broadcast scalar
    k.consume(sb, [amd.array.Scalar(None, amd.array.int64, False), gb.slice(100, 3)])  # null scalar
    k.merge(sa, sb, HostArray(perm, None, 0, nb).to_device(amd))
    out = k.finalize(sa)

    # oracle restatement of the very same call sequence
    oa, ob = O.HashSumState(skip_nulls, min_count), O.HashSumState(skip_nulls, min_count)
    oa.resize(num_groups)
    oa.consume(np.ascontiguousarray(vals.values), vals.valid_bitmap(), vals.offset, gid_a)
    ob.resize(nb)
    ob.consume(np.ascontiguousarray(vals_b.values), vals_b.valid_bitmap(), vals_b.offset, gid_b_local)
    ob.consume(None, None, 0, gid_b_local[:100], scalar=(7, True))
    ob.consume(None, None, 0, gid_b_local[100:103], scalar=(0, False))
    oa.merge(ob, perm)
    want_sums, want_valid, want_nulls = oa.finalize()

    tag = f"hash_sum_kernel[n={n},G={num_groups},skip_nulls={skip_nulls},min_count={min_count}]"
    assert out.length == num_groups and out.type == amd.array.int64
    got_valid, pad_ok = _logical_valid(out)
    assert pad_ok, tag + ": validity padding bits not zero"
    assert_equal(got_valid, want_valid, tag + " validity")
    assert_equal(_data_np(out, np.int64)[got_valid], want_sums[want_valid], tag + " sums")
    if skip_nulls:
        assert out.null_count == want_nulls, tag
        if want_nulls == 0:
            assert out.validity is None  # Finish only allocates a bitmap when a group is null
    if use_pyarrow and pa is not None:
        # the reference, end to end: same rows keyed by the global group id
        keys = np.concatenate([gid_a, perm[gid_b_local], perm[gid_b_local[:100]], perm[gid_b_local[100:103]]])
        allv = pa.concat_arrays([vals.to_pyarrow(), vals_b.to_pyarrow(),
                                 pa.array(np.full(100, 7, dtype=np.int64)),
                                 pa.array([None] * 3, type=pa.int64())])
        t = pa.table({"k": pa.array(keys.astype(np.int64)), "v": allv})
        r = t.group_by("k", use_threads=False).aggregate(
            [("v", "sum", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count))])
        rk = r.column("k").combine_chunks().to_numpy()
        rs = r.column("v_sum").combine_chunks()
        rvalid = ~np.asarray(rs.is_null())
        rsum = rs.fill_null(0).to_numpy(zero_copy_only=False)
        for kk, vv, ok in zip(rk.tolist(), rsum.tolist(), rvalid.tolist()):
            assert bool(got_valid[kk]) == ok, f"{tag}: group {kk} validity vs pyarrow"
            if ok:
                assert int(_data_np(out, np.int64)[kk]) == vv, f"{tag}: group {kk} sum vs pyarrow"
    return out


# ------------------------------------------------------------------ null_count bookkeeping
def check_null_count_bookkeeping(amd, rng, n=10_000):
    """the same logical array described in every way Arrow allows —
    exact null_count, kUnknownNullCount (-1), a validity buffer that is present but all-valid
    (null_count 0), and NO validity buffer with an unknown count (vector_selection_test.cc:625) —
    must filter / take identically, and the outputs must follow the reference's rules:
    filter: validity allocated iff an input may have nulls (:472), null_count 0 or unknown
    (:462-467); take: exact null_count (vector_selection_take_internal.cc:377)."""
    vals = util.random_array(rng, np.int64, n, null_p=0.2, offset=2)
    mask = util.random_mask(rng, n, 0.3, null_p=0.1, offset=1)
    idx = HostArray(rng.integers(0, n, 4000).astype(np.int32), rng.random(4000) >= 0.1, 0, 4000)
    for sel in ("drop", "emit_null"):
        ref_f = check_filter(amd, vals, mask, sel, use_pyarrow=False)
        ref_t = check_take(amd, vals, idx, boundscheck=True, use_pyarrow=False)
        base_v, base_m, base_i = vals.to_device(amd), mask.to_device(amd), idx.to_device(amd)
        unk = lambda a: amd.Array(a.type, a.length, a.buffers, -1, a.offset)  # noqa: E731
        out = amd.compute.filter(unk(base_v), unk(base_m), sel)
        assert out.length == ref_f.length
        assert_equal(_data_np(out, np.int64), _data_np(ref_f, np.int64), "filter with unknown null counts")
        assert_equal(_logical_valid(out)[0], _logical_valid(ref_f)[0], "filter validity with unknown null counts")
        assert out.null_count == -1 and out.validity is not None
        tk = amd.compute.take(unk(base_v), unk(base_i))
        assert_equal(_logical_valid(tk)[0], _logical_valid(ref_t)[0], "take validity with unknown null counts")
        assert tk.null_count == ref_t.null_count == int((~_logical_valid(ref_t)[0]).sum())
    # validity buffers present but all valid: null_count 0 => treated as no nulls, no output bitmap
    allv = util.random_array(rng, np.int64, n, offset=3)
    allm = util.random_mask(rng, n, 0.4)
    dv = HostArray(allv.values, np.ones(len(allv.values), bool), allv.offset, n).to_device(amd)
    dm = HostArray(allm.values, np.ones(len(allm.values), bool), allm.offset, n).to_device(amd)
    assert dv.validity is not None and dm.validity is not None
    dv0 = amd.Array(dv.type, dv.length, dv.buffers, 0, dv.offset)
    dm0 = amd.Array(dm.type, dm.length, dm.buffers, 0, dm.offset)
    out = amd.compute.filter(dv0, dm0, "emit_null")
    want, _ = O.filter(allv.data_bytes(), None, allv.offset, allm.data_bytes(), None, allm.offset, n, 1, True)
    assert out.validity is None and out.null_count == 0 and out.length == len(want)
    assert_equal(_data_np(out, np.int64), want, "filter of all-valid arrays that carry bitmaps")
    # ... and the same bitmaps with an UNKNOWN count: may have nulls => a bitmap comes out, all set
    out = amd.compute.filter(amd.Array(dv.type, dv.length, dv.buffers, -1, dv.offset), dm0, "drop")
    assert out.validity is not None and bool(_logical_valid(out)[0].all())
    assert_equal(_data_np(out, np.int64), O.filter(allv.data_bytes(), None, allv.offset, allm.data_bytes(), None,
                                                    allm.offset, n, 0, True)[0], "filter, unknown count, all valid")
    # no validity buffer + unknown null_count (vector_selection_test.cc:625): no nulls, no bitmap read
    nv = amd.Array(dv.type, dv.length, [None, dv.buffers[1]], -1, dv.offset)
    nm = amd.Array(dm.type, dm.length, [None, dm.buffers[1]], -1, dm.offset)
    out = amd.compute.filter(nv, nm, "emit_null")
    assert out.length == len(want)
    assert_equal(_data_np(out, np.int64), want, "filter without bitmaps but unknown null_count")
    tk = amd.compute.take(nv, amd.Array(base_i.type, 100, [None, base_i.buffers[1]], -1, 0))
    assert tk.validity is None and tk.null_count == 0
    assert_equal(_data_np(tk, np.int64), allv.logical_values()[idx.values[:100]], "take without bitmaps")


COPY_SEG = np.dtype([("src", "<u8"), ("dst", "<u8"), ("nbytes", "<u8")])


def check_copy_segments(amd, rng, scale=1):
    """arx_copy_segments: many device-to-device copies in one launch, every combination of source / destination
    alignment (the 16-byte path aligns the destination and reads the source wherever it is), lengths around the
    16-byte and 64 KiB steps, empty segments; bytes between the destinations stay untouched."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    lengths = [0, 1, 15, 16, 17, 31, 33, 255, 4096, 65535, 65536, 65537, 70001 * scale, 3 * 65536 + 5]
    segs = []
    src_pos = dst_pos = 0
    for i, n in enumerate(lengths * 2):
        src_pos += int(rng.integers(0, 16)) + (i % 5 == 0) * 16
        dst_pos += int(rng.integers(1, 16)) + 8          # a gap before every destination
        segs.append((src_pos, dst_pos, n))
        src_pos += n
        dst_pos += n
    src_h = rng.integers(0, 256, src_pos + 64, dtype=np.uint8)
    dst_h = rng.integers(0, 256, dst_pos + 64, dtype=np.uint8)
    src = to_device(src_h, dev)
    dst = to_device(dst_h.copy(), dev)
    table = np.zeros(len(segs), COPY_SEG)
    for i, (a, b, n) in enumerate(segs):
        table[i] = (src.data_ptr() + a, dst.data_ptr() + b, n)
    d_table = to_device(table.view(np.uint8), dev)
    _lib.check(lib.arx_copy_segments(d_table.data_ptr(), len(segs), max(n for _, _, n in segs), current_stream(dev)))
    want = dst_h.copy()
    for a, b, n in segs:
        want[b:b + n] = src_h[a:a + n]
    assert_equal(dst.cpu().numpy()[: len(want)], want, "copy_segments")


BIT_SEG = np.dtype([("src", "<u8"), ("src_bit_offset", "<i8"), ("dst", "<u8"), ("dst_bit_offset", "<i8"), ("nbits", "<i8")])


def check_buffer_copy(amd, rng, scale=1):
    """arx_buffer_copy: byte ranges of every length around the 16-byte / 4-KiB boundaries, source and destination at
    equal and at different 16-byte phases; bytes outside the destination range stay untouched."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    big = 3_000_017 * scale
    src_h = rng.integers(0, 256, big + 64, dtype=np.uint8)
    src = to_device(src_h, dev)
    for n in [0, 1, 15, 16, 17, 255, 4095, 4096, 4097, 65536 + 3, big]:
        for so, do in [(0, 0), (5, 5), (3, 19), (1, 2), (16, 7)]:
            if so + n > len(src_h):
                continue
            dst = torch.full((n + 64,), 0xA5, dtype=torch.uint8, device=dev)
            _lib.check(lib.arx_buffer_copy(src.data_ptr() + so, dst.data_ptr() + do, n, current_stream(dev)))
            got = dst.cpu().numpy()
            want = np.full(n + 64, 0xA5, np.uint8)
            want[do:do + n] = src_h[so:so + n]
            assert_equal(got, want, f"buffer_copy n={n} src+{so} dst+{do}")
    out = amd.compute.copy_buffer(src)
    assert_equal(out.cpu().numpy()[: len(src_h)], src_h, "copy_buffer")


def check_bytes_to_bitmap(amd, rng, scale=1):
    """arx_bytes_to_bitmap (BytesToBits, util/bitmap_builders.cc): one byte per row -> LSB-first bitmap, zero padding in the
    last word, the set-bit count ADDED to a device counter; lengths around the 64-row word, source at odd addresses."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    for n in [0, 1, 63, 64, 65, 4095, 4097, 100_003 * scale]:
        for shift, p in [(0, 0.5), (3, 0.02), (1, 1.0)]:
            host = (rng.random(n + shift) < p).astype(np.uint8) * rng.integers(1, 256, n + shift, dtype=np.uint8)   # any non-zero byte is "set"
            src = to_device(host, dev)
            out = torch.full(((n + 63) // 64 * 8 + 16,), 0xA5, dtype=torch.uint8, device=dev)
            counter = torch.full((1,), 1000, dtype=torch.int64, device=dev)
            _lib.check(lib.arx_bytes_to_bitmap(src.data_ptr() + shift, n, out.data_ptr(), counter.data_ptr(), current_stream(dev)))
            _lib.check(lib.arx_bytes_to_bitmap(src.data_ptr() + shift, n, out.data_ptr(), None, current_stream(dev)))   # no counter
            want_bits = host[shift:] != 0
            want = np.packbits(np.concatenate([want_bits, np.zeros((-n) % 64, bool)]), bitorder="little")
            got = out.cpu().numpy()
            assert_equal(got[: len(want)], want, f"bytes_to_bitmap n={n} shift={shift}")
            assert (got[len(want):] == 0xA5).all(), "wrote past the last word"
            assert int(counter.item()) == 1000 + int(want_bits.sum()), (n, shift)


def check_hash_any_all_kernels(amd, rng, n=6000, num_groups=41, null_p=0.2):
    """hash_any / hash_all as the plugin runs them: three dense counts per group (valid rows, null rows, valid AND true
    rows — hash_count kernels, the third over arx_bitmap_and of validity and values), merged through a group_id_mapping,
    finalized by arx_hash_bool_finalize; against the oracle's restatement of GroupedBooleanAggregator and, end to end,
    against pyarrow's own hash_any / hash_all on the same rows; skip_nulls on / off, min_count 1 / 3, offsets."""
    import pyarrow as pa
    import pyarrow.compute as pc
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device
    from oracle import oracle as O

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    for off, p_true in ((0, 0.97), (5, 0.03), (3, 0.5)):
        vals = rng.random(n + off) < p_true
        ok = rng.random(n + off) >= null_p
        gids = rng.integers(0, num_groups, n).astype(np.uint32)
        gids[gids == 7] = 8                                    # group 7 stays empty
        ok[off:][gids == 3] = False                            # group 3 sees only nulls
        bits = lambda b: np.packbits(np.concatenate([b, np.zeros((-len(b)) % 64, bool)]), bitorder="little")
        d_vals, d_valid, d_gids = to_device(bits(vals), dev), to_device(bits(ok), dev), to_device(gids, dev)
        halves = [(0, n // 3), (n // 3, n)]                    # two states, merged through a mapping
        for skip_nulls, min_count in ((True, 1), (False, 1), (True, 3), (False, 3)):
            for is_all in (False, True):
                states, oracles = [], []
                for lo, hi in halves:
                    cnt = [torch.zeros(num_groups, dtype=torch.int64, device=dev) for _ in range(3)]
                    m = hi - lo
                    both = torch.zeros((m + 63) // 64 * 8 + 16, dtype=torch.uint8, device=dev)
                    g_ptr = d_gids.data_ptr() + lo * 4
                    _lib.check(lib.arx_hash_count_consume(d_valid.data_ptr(), off + lo, -1, 0, g_ptr, m, cnt[0].data_ptr(), st))
                    _lib.check(lib.arx_hash_count_consume(d_valid.data_ptr(), off + lo, -1, 1, g_ptr, m, cnt[1].data_ptr(), st))
                    _lib.check(lib.arx_bitmap_and(d_valid.data_ptr(), off + lo, d_vals.data_ptr(), off + lo, m, both.data_ptr(), st))
                    _lib.check(lib.arx_hash_count_consume(both.data_ptr(), 0, -1, 0, g_ptr, m, cnt[2].data_ptr(), st))
                    states.append(cnt)
                    o = O.HashBoolState(is_all, skip_nulls, min_count)
                    o.resize(num_groups)
                    o.consume(vals[off + lo:off + hi], bits(ok), off + lo, gids[lo:hi])
                    oracles.append(o)
                mapping = rng.permutation(num_groups).astype(np.uint32)
                d_map = to_device(mapping, dev)
                for a, b in zip(states[0], states[1]):
                    _lib.check(lib.arx_hash_count_merge(a.data_ptr(), b.data_ptr(), d_map.data_ptr(), num_groups, st))
                oracles[0].merge(oracles[1], mapping)
                words = (num_groups + 63) // 64
                out_v = torch.full((words * 8 + 8,), 0xA5, dtype=torch.uint8, device=dev)
                out_ok = torch.full((words * 8 + 8,), 0xA5, dtype=torch.uint8, device=dev)
                counter = torch.full((1,), 100, dtype=torch.int64, device=dev)
                nv, nn, nt = states[0]
                _lib.check(lib.arx_hash_bool_finalize(nv.data_ptr(), nn.data_ptr(), nt.data_ptr(), num_groups, int(is_all),
                                                      int(skip_nulls), min_count, out_v.data_ptr(), out_ok.data_ptr(),
                                                      counter.data_ptr(), st))
                got_v = np.unpackbits(out_v.cpu().numpy()[: words * 8], bitorder="little")[:num_groups].astype(bool)
                got_ok = np.unpackbits(out_ok.cpu().numpy()[: words * 8], bitorder="little")[:num_groups].astype(bool)
                want_v, want_ok = oracles[0].finalize()
                tag = f"hash_{'all' if is_all else 'any'} skip_nulls={skip_nulls} min_count={min_count} off={off}"
                assert_equal(got_ok, want_ok, tag + " validity")
                assert_equal(got_v[want_ok], want_v[want_ok], tag + " values")
                assert int(counter.item()) == 100 + int(want_ok.sum()), tag
                assert (out_v.cpu().numpy()[words * 8:] == 0xA5).all() and (out_ok.cpu().numpy()[words * 8:] == 0xA5).all()
        # end to end against pyarrow itself (one state over all rows: rows = state 1's mapping applied to nothing)
        col = pa.array(vals[off:], mask=~ok[off:])
        tab = pa.table({"g": pa.array(gids), "b": col})
        for skip_nulls, min_count in ((True, 1), (False, 2)):
            opts = pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count)
            ref = tab.group_by("g", use_threads=False).aggregate([("b", "any", opts), ("b", "all", opts)]).sort_by("g")
            for is_all, name in ((False, "b_any"), (True, "b_all")):
                o = O.HashBoolState(is_all, skip_nulls, min_count)
                o.resize(num_groups)
                o.consume(vals[off:], bits(ok), off, gids)
                v_, ok_ = o.finalize()
                seen = np.asarray(ref.column("g"))
                want = ref.column(name).to_pylist()
                got = [bool(v_[g]) if ok_[g] else None for g in seen]
                assert got == want, (name, skip_nulls, min_count)


def check_groupby_key_range(amd, rng, scale=1):
    """arx_groupby_key_range_i32: {min, max} of an int32 key column folded into the caller's pair (atomic min / max), at
    offsets, for empty, tiny and large columns and keys at both ends of the int32 range."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    for n, lo, hi in [(0, 0, 1), (1, -5, -4), (63, 0, 10), (65, -2**31, 2**31), (1000, 7, 8), (200_003 * scale, -123456, 9_876_543)]:
        for off in (0, 5):
            host = rng.integers(lo, hi, n + off, dtype=np.int64).astype(np.int32)
            keys = to_device(host, dev)
            pair = torch.tensor([2**31 - 1, -2**31], dtype=torch.int32, device=dev)
            span = _lib.ArxSpan(None, keys.data_ptr(), off, n, 0)
            _lib.check(lib.arx_groupby_key_range_i32(span, pair.data_ptr(), current_stream(dev)))
            got = pair.cpu().numpy()
            if n == 0:
                assert_equal(got, np.array([2**31 - 1, -2**31], np.int32), "key range of an empty column: untouched")
            else:
                assert_equal(got, np.array([host[off:].min(), host[off:].max()], np.int32), f"key range n={n} off={off}")
            # folding: a second column only widens the pair
            if n > 0:
                g0, g1 = int(got[0]), int(got[1])
                inner = np.array([g0 + 1, g1 - 1, g1], np.int32) if g1 - g0 > 1 else host[off:off + 1]
                more = to_device(inner, dev)      # (a padded byte buffer: the row count is inner's)
                span2 = _lib.ArxSpan(None, more.data_ptr(), 0, len(inner), 0)
                _lib.check(lib.arx_groupby_key_range_i32(span2, pair.data_ptr(), current_stream(dev)))
                assert_equal(pair.cpu().numpy(), got, "a narrower column must not move the pair")


SORT_RECORD = np.dtype([("key_lo", "<u4"), ("key_hi", "<u4"), ("row", "<u4")])


def check_sort_records(amd, rng, n, options=(), ties=True):
    """arx_sort_records: 12-byte {key, row} records in NO particular order -> their rows by (key, row) ascending (what the
    receiver of the sharded sort's records form runs; = the order a stable sort of the keys gives when the rows are the
    global row numbers).  Against numpy's lexsort on the same records: keys with many ties, rows a shuffled run of
    distinct 32-bit numbers starting high (so that a kernel that took them for positions would fault or misorder)."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import alloc, current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    for k_, v_ in options:
        assert lib.arx_set_option(k_, v_) == 0, k_
    try:
        keys = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        if ties:
            keys[::3] = keys[::3] % np.uint64(50)
            keys[1::7] = keys[(1 + 7 * (np.arange(len(keys[1::7])) // 2 * 2)) % n]      # exact duplicates of other rows' keys
        rows = (np.uint64(2**32 - 1 - n) + rng.permutation(n).astype(np.uint64)).astype(np.uint32)
        rec = np.empty(n, SORT_RECORD)
        rec["key_lo"], rec["key_hi"], rec["row"] = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), (keys >> np.uint64(32)).astype(np.uint32), rows
        big = n > 20_000_000      # (numpy's lexsort of 1.5e8 pairs takes a minute of the GPU gate: beyond 2e7 the order is checked on the device)
        want = None if big else rows[np.lexsort((rows, keys))].astype(np.uint64)
        drec = to_device(rec.view(np.uint8), dev)
        out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        ws_bytes = lib.arx_sort_indices_workspace_bytes(n) + 256
        ws = alloc(ws_bytes, dev)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        _lib.check(lib.arx_sort_records(drec.data_ptr(), n, ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), out.data_ptr(), current_stream(dev)))
        if big:
            # the rows come back as a permutation of the records' rows, and (key, row) never decreases along it
            base = int(2**32 - 1 - n)
            pos = torch.empty(n, dtype=torch.int64, device=dev)
            pos[torch.from_numpy((rows.astype(np.int64) - base)).to(dev)] = torch.arange(n, dtype=torch.int64, device=dev)   # row -> record
            got_rows = out[:n]
            seen = torch.zeros(n, dtype=torch.bool, device=dev)
            seen[got_rows - base] = True
            assert bool(seen.all()), "arx_sort_records: the output is not a permutation of the rows"
            ks = torch.from_numpy(keys.view(np.int64)).to(dev)[pos[got_rows - base]] ^ torch.iinfo(torch.int64).min   # unsigned order as signed
            ok = (ks[1:] > ks[:-1]) | ((ks[1:] == ks[:-1]) & (got_rows[1:] > got_rows[:-1]))
            assert bool(ok.all()), f"arx_sort_records n={n}: (key, row) decreases at {int((~ok).nonzero()[0])}"
        else:
            got = out[:n].cpu().numpy().view(np.uint64)
            assert_equal(got, want, f"arx_sort_records n={n} options={options}")
    finally:
        for k_, v_ in options:
            lib.arx_set_option(k_, {b"sort_msd": -1, b"sort_msd_sampled": 1, b"sort_msd_segment_rows": 1 << 27, b"sort_msd_wide": 1,
                                    b"sort_msd_wide_bits": 0}.get(k_, 0))


def _gbl_sampled_unit(s, stride):
    """gbl_sampled_unit of arrow_amd/csrc/groupby_lines.h (the unit of stratum s that the lines plan's samples read)."""
    m = (1 << 64) - 1
    z = (s * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z ^= z >> 29
    return s * stride + z % stride


def check_groupby_lines_plan(amd, rng_for, scale=1, wide_width=True):
    """The LINES plan of the partitioned consume (groupby_lines.h: partitions = slices of the key range, a scatter whose
    every store is a whole 128-byte line, direct-indexed LDS tables) forced on at small sizes, every case against the
    oracle: one-shot and in several consumes, nulls in keys and values at offsets, negative ranges, a full and a
    sampled histogram, rows outside the sampled range (their own pass), a hot key (rounds without end: the plan gives
    the rows back untouched), a range too wide or too narrow (declined), the 12288-wide partitions.  The counters say
    which way every case went."""
    lib = amd._lib.get_lib()
    ctr = lambda name: int(lib.arx_get_counter(name))      # noqa: E731
    knobs = {b"groupby_partition_min_rows": 0, b"groupby_lines_min_rows": 1, b"groupby_lines_wgs": 2,
             b"groupby_lines_unit_rows": 4096, b"groupby_lines_sample_rows": 1 << 24}
    for k_, v_ in knobs.items():
        assert lib.arx_set_option(k_, v_) == 0, k_
    try:
        n = 30000 * scale
        # 1. uniform ids, no nulls, one consume: the plan runs (full histogram: n < the sample)
        s0, f0, d0 = ctr(b"groupby_slices_lines"), ctr(b"groupby_lines_fallbacks"), ctr(b"groupby_lines_declined")
        rng = rng_for("gbl", 1)
        k = util.random_array(rng, np.int32, n, lo=0, hi=50000)
        v = util.random_array(rng, np.int64, n)
        check_groupby_sum(amd, k, v, use_pyarrow=False)
        assert ctr(b"groupby_slices_lines") == s0 + 1 and ctr(b"groupby_lines_fallbacks") == f0, "uniform ids must take the lines plan"
        # 2. nulls in keys and values, offsets, a range around zero, two consumes into one table, both null options
        for skip_nulls, min_count in ((True, 1), (False, 2)):
            rng = rng_for("gbl", 2, skip_nulls)
            k = util.random_array(rng, np.int32, n, null_p=0.03, offset=3, lo=-30000, hi=30000)
            v = util.random_array(rng, np.int64, n, null_p=0.1, offset=1)
            s1 = ctr(b"groupby_slices_lines")
            check_groupby_sum(amd, k, v, skip_nulls=skip_nulls, min_count=min_count, batches=2, use_pyarrow=skip_nulls)
            assert ctr(b"groupby_slices_lines") == s1 + 2
        # 3. a sampled histogram (one 64-row unit in 8) and rows outside the sampled range: keys far away in a unit the
        #    sample does not read are counted by the scatter and consumed by their own pass
        assert lib.arx_set_option(b"groupby_lines_sample_rows", max(64, n // 8)) == 0
        assert lib.arx_set_option(b"groupby_lines_range_sample_rows", max(64, n // 8)) == 0
        stride = max(1, n // max(64, n // 8))
        assert stride > 1
        sampled = {_gbl_sampled_unit(s, stride) for s in range((n + 63) // 64 // stride + 2)}
        unit = next(u for u in range(5, n // 64 - 1) if u not in sampled)
        for null_p in (0.0, 0.05):
            rng = rng_for("gbl", 3, null_p)
            k = util.random_array(rng, np.int32, n, null_p=null_p, lo=1000, hi=200000)
            v = util.random_array(rng, np.int64, n, null_p=null_p)
            far = np.array([2**30 + 7, -2**31, 2**31 - 1, 2**30 + 7, -5], np.int32)
            k.values[k.offset + unit * 64 + 3: k.offset + unit * 64 + 3 + len(far)] = far
            s1, o1 = ctr(b"groupby_slices_lines"), ctr(b"groupby_lines_outlier_rows")
            check_groupby_sum(amd, k, v, use_pyarrow=False)
            assert ctr(b"groupby_slices_lines") == s1 + 1
            assert ctr(b"groupby_lines_outlier_rows") - o1 >= (3 if null_p else 4), "the far keys are outside the sampled range"
        assert lib.arx_set_option(b"groupby_lines_sample_rows", 1 << 24) == 0
        assert lib.arx_set_option(b"groupby_lines_range_sample_rows", 1 << 20) == 0
        # 4. a hot key: the scatter would need hundreds of rounds per batch — it gives up, nothing consumed, the other plans run
        rng = rng_for("gbl", 4)
        k = util.random_array(rng, np.int32, n, lo=0, hi=50000)
        k.values[rng.random(len(k.values)) < 0.9] = 4242
        v = util.random_array(rng, np.int64, n)
        f1, s1 = ctr(b"groupby_lines_fallbacks"), ctr(b"groupby_slices_lines")
        check_groupby_sum(amd, k, v, use_pyarrow=False)
        assert ctr(b"groupby_lines_fallbacks") == f1 + 1 and ctr(b"groupby_slices_lines") == s1, "a hot key must send the rows to the other plans"
        # 5. declined before anything runs: keys over the whole int32 range, and a range of a few keys
        for lo, hi in ((-2**31, 2**31 - 1), (10, 2500)):
            rng = rng_for("gbl", 5, lo)
            k = util.random_array(rng, np.int32, n // 2, lo=lo, hi=hi)
            v = util.random_array(rng, np.int64, n // 2)
            d1, s1 = ctr(b"groupby_lines_declined"), ctr(b"groupby_slices_lines")
            check_groupby_sum(amd, k, v, use_pyarrow=False)
            assert ctr(b"groupby_lines_declined") == d1 + 1 and ctr(b"groupby_slices_lines") == s1
        # 6. partitions of 12288 keys (a range beyond 1216 x 8192) and of 8192
        for hi in ((12_000_000, 9_500_000) if wide_width else ()):
            rng = rng_for("gbl", 6, hi)
            k = util.random_array(rng, np.int32, n, null_p=0.01, lo=-1000, hi=hi)
            v = util.random_array(rng, np.int64, n)
            s1 = ctr(b"groupby_slices_lines")
            check_groupby_sum(amd, k, v, use_pyarrow=False)
            assert ctr(b"groupby_slices_lines") == s1 + 1
        # 7. wrap-around sums inside a few groups of a dense range
        rng = rng_for("gbl", 7)
        k = util.random_array(rng, np.int32, n, lo=100, hi=7000)
        v = util.random_array(rng, np.int64, n, lo=2**62, hi=2**63 - 1)
        check_groupby_sum(amd, k, v, use_pyarrow=False)
        assert d0 <= ctr(b"groupby_lines_declined")
    finally:
        for k_, v_ in {b"groupby_partition_min_rows": 1 << 17, b"groupby_lines_min_rows": 1 << 22, b"groupby_lines_wgs": 0,
                       b"groupby_lines_unit_rows": 1 << 21, b"groupby_lines_sample_rows": 1 << 24,
                       b"groupby_lines_range_sample_rows": 1 << 20}.items():
            lib.arx_set_option(k_, v_)


def check_groupby_range_state(amd, rng_for, scale=1):
    """The range-partitioned group-by STATE through its C ABI (arx_groupby_range_*; compute.RangeGroupBySum): plan ->
    consume (several batches into one state) -> finalize, and two states merged by partition runs the way two ranks
    exchange them -> finalize per run; every result against the oracle on the same rows, in ascending key order.  What
    the state declines (nulls, a key outside the plan, a hot key, a range too wide or too narrow) is declined with
    nothing consumed."""
    import torch

    R = amd.compute.RangeGroupBySum
    lib = amd._lib.get_lib()
    for k_, v_ in {b"groupby_lines_wgs": 2, b"groupby_lines_unit_rows": 4096}.items():
        assert lib.arx_set_option(k_, v_) == 0
    try:
        n = 25000 * scale
        for lo, hi, min_count in ((0, 40000, 1), (-70000, -20000, 3), (5, 9_700_000 if scale > 1 else 300_000, 1)):
            rng = rng_for("range-state", lo, hi)
            k = util.random_array(rng, np.int32, n, offset=3, lo=lo, hi=hi)
            v = util.random_array(rng, np.int64, n, offset=1)
            plan = R.plan_for(n, lo, hi, sampled=False)
            assert plan is not None and plan.key_min == lo and plan.slots >= hi - lo + 1
            dk, dv = k.to_device(amd), v.to_device(amd)
            # the sampled range the sharded path starts from brackets nothing outside the true one
            neg_lo, smax = [int(x) for x in R.sampled_key_range(dk, 1 << 30).cpu().tolist()]
            kv = k.values[k.offset:k.offset + n]
            assert -neg_lo == int(kv.min()) and smax == int(kv.max())
            opts = amd.compute.ScalarAggregateOptions(True, min_count)
            st = R(plan, dk.device, opts)
            for b in range(0, n, (n + 2) // 3):        # three consumes into one state
                m = min((n + 2) // 3, n - b)
                assert st.consume(dk.slice(b, m), dv.slice(b, m))
            gk, gkv, gs, gvalid = st.finalize()
            w = O.groupby_sum_i64(np.ascontiguousarray(k.values), None, k.offset, np.ascontiguousarray(v.values), None, v.offset,
                                  n, True, min_count)
            order = np.argsort(w["keys"], kind="stable")
            assert_equal(gk.cpu().numpy(), w["keys"][order], f"range state keys [{lo}, {hi}]")
            assert_equal(gs.cpu().numpy(), w["sums"][order], "range state sums")
            assert_equal(gvalid.cpu().numpy().astype(bool), w["valid"][order].astype(bool), "range state validity (min_count)")
            assert bool(gkv.all())
            # two states (two "ranks": the halves of the rows), merged run by run as the sharded path does
            a, b2 = R(plan, dk.device, opts), R(plan, dk.device, opts)
            assert a.consume(dk.slice(0, n // 2), dv.slice(0, n // 2)) and b2.consume(dk.slice(n // 2, n - n // 2), dv.slice(n // 2, n - n // 2))
            parts, pb = int(plan.partitions), a.partition_bytes()
            got_k, got_s = [], []
            for first, cnt in ((0, parts // 3), (parts // 3, parts - parts // 3)):
                blocks = torch.cat([a.state[first * pb // 8:(first + cnt) * pb // 8], b2.state[first * pb // 8:(first + cnt) * pb // 8]])
                amd._lib.check(lib.arx_groupby_range_merge(blocks.data_ptr(), blocks.data_ptr() + cnt * pb, plan.width, cnt, 1, cnt * pb,
                                                           amd.array.current_stream(dk.device)))
                fk, _, fs, _ = a.finalize(first, cnt, blocks=blocks)
                got_k.append(fk.cpu().numpy())
                got_s.append(fs.cpu().numpy())
            assert_equal(np.concatenate(got_k), w["keys"][order], "two merged range states: keys")
            assert_equal(np.concatenate(got_s), w["sums"][order], "two merged range states: sums")
        # declined, nothing consumed: a key outside the plan; a hot key; rows with nulls
        rng = rng_for("range-state", "declines")
        plan = R.plan_for(n, 0, 50000, sampled=False)
        st = R(plan, None)
        k = util.random_array(rng, np.int32, n, lo=0, hi=50000)
        v = util.random_array(rng, np.int64, n)
        k.values[n // 2] = 50000 + int(plan.slots)
        assert not st.consume(k.to_device(amd), v.to_device(amd))
        k.values[:] = 17
        assert not st.consume(k.to_device(amd), v.to_device(amd))
        kn = util.random_array(rng, np.int32, n, null_p=0.1, lo=0, hi=50000)
        assert not st.consume(kn.to_device(amd), v.to_device(amd))
        assert int(st.state.abs().sum()) == 0, "a declined consume must leave the state untouched"
        assert len(st.finalize()[0]) == 0
        # ranges the state does not plan for
        assert R.plan_for(n, 0, 3000, sampled=False) is None and R.plan_for(n, -2**31, 2**31 - 1, sampled=False) is None
        assert R.plan_for(n, 0, 3071, sampled=False).width == 8
        assert R.plan_for(n, 0, 1216 * 12288 - 1, sampled=False) is not None and R.plan_for(n, 0, 1216 * 12288, sampled=False) is None
    finally:
        lib.arx_set_option(b"groupby_lines_wgs", 0)
        lib.arx_set_option(b"groupby_lines_unit_rows", 1 << 21)


def check_hash_minmax_count_kernels(amd, rng, n=5000, num_groups=37, null_p=0.2):
    """The dense-id state kernels behind hash_min / hash_max / hash_min_max / hash_count, driven through the C ABI the
    way GroupByNode drives a HashAggregateKernel: two states, resize (fill), consume (arrays at offsets, a broadcast
    scalar, a null scalar), merge through a group_id_mapping, finalize; against the oracle's restatement and, end to
    end, against pyarrow's own hash_min_max / hash_count on the same rows."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    vals = util.random_array(rng, np.int64, n, null_p=null_p, offset=3)
    vals_b = util.random_array(rng, np.int64, n, null_p=null_p, offset=0)
    gid_a = rng.integers(0, num_groups, size=n).astype(np.uint32)
    nb = max(1, num_groups // 2)
    perm = rng.permutation(num_groups)[:nb].astype(np.uint32)
    gid_b = rng.integers(0, nb, size=n).astype(np.uint32)
    dva, dvb = vals.to_device(amd), vals_b.to_device(amd)
    dga, dgb, dperm = to_device(gid_a, dev), to_device(gid_b, dev), to_device(perm, dev)

    def state(g):
        mins = torch.zeros(g, dtype=torch.int64, device=dev)
        maxs = torch.zeros(g, dtype=torch.int64, device=dev)
        seen = torch.zeros(g, dtype=torch.int32, device=dev)
        return mins, maxs, seen

    half = num_groups // 2 + 1
    a_mins, a_maxs, a_seen = state(num_groups)
    _lib.check(lib.arx_hash_minmax_i64_fill(a_mins.data_ptr(), a_maxs.data_ptr(), 0, min(half, num_groups), st))    # Resize in
    _lib.check(lib.arx_hash_minmax_i64_fill(a_mins.data_ptr(), a_maxs.data_ptr(), min(half, num_groups),             # two steps
                                            num_groups - min(half, num_groups), st))
    b_mins, b_maxs, b_seen = state(nb)
    _lib.check(lib.arx_hash_minmax_i64_fill(b_mins.data_ptr(), b_maxs.data_ptr(), 0, nb, st))
    sp_a, sp_b = dva.span(), dvb.span()
    _lib.check(lib.arx_hash_minmax_i64_consume(C.byref(sp_a), 0, 0, dga.data_ptr(), n, a_mins.data_ptr(), a_maxs.data_ptr(),
                                               a_seen.data_ptr(), st))
    _lib.check(lib.arx_hash_minmax_i64_consume(C.byref(sp_b), 0, 0, dgb.data_ptr(), n, b_mins.data_ptr(), b_maxs.data_ptr(),
                                               b_seen.data_ptr(), st))
    scal = _lib.ArxSpan(None, None, 0, 100, 0)
    _lib.check(lib.arx_hash_minmax_i64_consume(C.byref(scal), 1, -7, dgb.data_ptr(), min(100, n), b_mins.data_ptr(),
                                               b_maxs.data_ptr(), b_seen.data_ptr(), st))
    nul = _lib.ArxSpan(None, None, 0, 3, 3)
    _lib.check(lib.arx_hash_minmax_i64_consume(C.byref(nul), 1, 0, dgb.data_ptr() + 4 * min(100, n - 3), 3, b_mins.data_ptr(),
                                               b_maxs.data_ptr(), b_seen.data_ptr(), st))
    _lib.check(lib.arx_hash_minmax_i64_merge(a_mins.data_ptr(), a_maxs.data_ptr(), a_seen.data_ptr(), b_mins.data_ptr(),
                                             b_maxs.data_ptr(), b_seen.data_ptr(), dperm.data_ptr(), nb, st))
    oa, ob = O.HashMinMaxState(), O.HashMinMaxState()
    oa.resize(num_groups)
    ob.resize(nb)
    oa.consume(vals.values, vals.valid_bitmap(), vals.offset, gid_a)
    ob.consume(vals_b.values, vals_b.valid_bitmap(), vals_b.offset, gid_b)
    ob.consume(None, None, 0, gid_b[:100], scalar=(-7, True))
    ob.consume(None, None, 0, gid_b[min(100, n - 3):min(100, n - 3) + 3], scalar=(0, False))
    oa.merge(ob, perm)
    for skip_nulls in (True, False):
        oa.skip_nulls = skip_nulls
        wmin, wmax, wvalid = oa.finalize()
        bits = torch.zeros((num_groups + 63) // 64, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(lib.arx_hash_minmax_i64_finalize(a_mins.data_ptr(), a_maxs.data_ptr(), a_seen.data_ptr(), num_groups,
                                                    int(skip_nulls), bits.data_ptr(), cnt.data_ptr(), st))
        got_valid = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:num_groups].astype(bool)
        tag = f"hash_minmax[n={n},G={num_groups},skip_nulls={skip_nulls}]"
        assert_equal(got_valid, wvalid, tag + " validity")
        assert int(cnt.item()) == int(wvalid.sum()), tag + " valid count"
        assert_equal(a_mins.cpu().numpy()[wvalid], wmin[wvalid], tag + " mins")
        assert_equal(a_maxs.cpu().numpy()[wvalid], wmax[wvalid], tag + " maxs")
    # ---- hash_count, the three modes
    for mode, name in ((0, "only_valid"), (1, "only_null"), (2, "all")):
        ca = torch.zeros(num_groups, dtype=torch.int64, device=dev)
        cb = torch.zeros(nb, dtype=torch.int64, device=dev)
        va = dva.validity.data_ptr() if dva.validity is not None and dva.null_count != 0 else None
        vb = dvb.validity.data_ptr() if dvb.validity is not None and dvb.null_count != 0 else None
        _lib.check(lib.arx_hash_count_consume(va, dva.offset, dva.null_count, mode, dga.data_ptr(), n, ca.data_ptr(), st))
        _lib.check(lib.arx_hash_count_consume(vb, dvb.offset, dvb.null_count, mode, dgb.data_ptr(), n, cb.data_ptr(), st))
        _lib.check(lib.arx_hash_count_consume(None, 0, 5, mode, dgb.data_ptr(), min(5, n), cb.data_ptr(), st))   # null scalar
        _lib.check(lib.arx_hash_count_consume(None, 0, 0, mode, dgb.data_ptr(), min(7, n), cb.data_ptr(), st))   # valid scalar
        _lib.check(lib.arx_hash_count_merge(ca.data_ptr(), cb.data_ptr(), dperm.data_ptr(), nb, st))
        ha, hb = O.HashCountState(name), O.HashCountState(name)
        ha.resize(num_groups)
        hb.resize(nb)
        ha.consume(vals.valid_bitmap(), vals.offset, gid_a)
        hb.consume(vals_b.valid_bitmap(), vals_b.offset, gid_b)
        hb.consume(None, 0, gid_b[:5], scalar_valid=False)
        hb.consume(None, 0, gid_b[:7], scalar_valid=True)
        ha.merge(hb, perm)
        assert_equal(ca.cpu().numpy(), ha.counts, f"hash_count[{name},n={n},G={num_groups}]")
    # ---- the oracle itself against the reference (pyarrow's hash kernels on the rows of state A)
    if pa is not None and n > 0:
        o1 = O.HashMinMaxState()
        o1.resize(num_groups)
        o1.consume(vals.values, vals.valid_bitmap(), vals.offset, gid_a)
        t = pa.table({"g": pa.array(gid_a), "v": vals.to_pyarrow()})
        r = t.group_by("g", use_threads=False).aggregate([("v", "min"), ("v", "max"), ("v", "count")]).sort_by("g")
        gs = r.column("g").to_numpy()
        mn, mx, valid = o1.finalize()
        assert_equal(valid[gs], ~np.asarray(r.column("v_min").is_null()), "oracle hash_min validity vs pyarrow")
        sel = valid[gs]
        assert_equal(mn[gs][sel], r.column("v_min").drop_null().to_numpy(), "oracle hash_min vs pyarrow")
        assert_equal(mx[gs][sel], r.column("v_max").drop_null().to_numpy(), "oracle hash_max vs pyarrow")
        h1 = O.HashCountState("only_valid")
        h1.resize(num_groups)
        h1.consume(vals.valid_bitmap(), vals.offset, gid_a)
        assert_equal(h1.counts[gs], r.column("v_count").to_numpy(), "oracle hash_count vs pyarrow")


def check_hash_minmax_float_kernels(amd, rng, dtype=np.float64, n=5000, num_groups=37, null_p=0.2):
    """arx_hash_minmax_float_consume / _finalize (float32 / float64 extrema as order keys in the int64 state arrays of
    the integer kernels), driven like a HashAggregateKernel: two states, fill, consume (arrays at offsets, a broadcast
    scalar, a null scalar, a NaN scalar), merge through a mapping, finalize — against the oracle's fmin / fmax
    restatement (NaNs skipped, groups of NaNs end as NaN, zeros of either sign compare equal) and against pyarrow."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    num_type = 9 if np.dtype(dtype) == np.float64 else 8
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, np.finfo(dtype).max, np.finfo(dtype).tiny, -np.finfo(dtype).tiny * 0.5], dtype)

    def column(offset):
        a = util.random_array(rng, dtype, n, null_p=null_p, offset=offset)
        v = a.values
        v *= (10.0 ** rng.integers(-30, 30, len(v))).astype(dtype)
        hit = rng.random(len(v)) < 0.3
        v[hit] = special[rng.integers(0, len(special), int(hit.sum()))]
        return a

    vals, vals_b = column(3), column(0)
    gid_a = rng.integers(0, num_groups, size=n).astype(np.uint32)
    if num_groups > 3 and n > 0:
        vals.values[vals.offset:vals.offset + n][gid_a == 1] = np.nan      # a group of NaNs only
    nb = max(1, num_groups // 2)
    perm = rng.permutation(num_groups)[:nb].astype(np.uint32)
    gid_b = rng.integers(0, nb, size=n).astype(np.uint32)
    dva, dvb = vals.to_device(amd), vals_b.to_device(amd)
    dga, dgb, dperm = to_device(gid_a, dev), to_device(gid_b, dev), to_device(perm, dev)

    def state(g):
        mins = torch.zeros(g, dtype=torch.int64, device=dev)
        maxs = torch.zeros(g, dtype=torch.int64, device=dev)
        seen = torch.zeros(g, dtype=torch.int32, device=dev)
        _lib.check(lib.arx_hash_minmax_i64_fill(mins.data_ptr(), maxs.data_ptr(), 0, g, st))
        return mins, maxs, seen

    a_mins, a_maxs, a_seen = state(num_groups)
    b_mins, b_maxs, b_seen = state(nb)
    sp_a, sp_b = dva.span(), dvb.span()
    _lib.check(lib.arx_hash_minmax_float_consume(C.byref(sp_a), num_type, 0, 0.0, dga.data_ptr(), n, a_mins.data_ptr(),
                                                 a_maxs.data_ptr(), a_seen.data_ptr(), st))
    _lib.check(lib.arx_hash_minmax_float_consume(C.byref(sp_b), num_type, 0, 0.0, dgb.data_ptr(), n, b_mins.data_ptr(),
                                                 b_maxs.data_ptr(), b_seen.data_ptr(), st))
    k = min(100, n)
    scal = _lib.ArxSpan(None, None, 0, k, 0)
    _lib.check(lib.arx_hash_minmax_float_consume(C.byref(scal), num_type, 1, -7.5, dgb.data_ptr(), k, b_mins.data_ptr(),
                                                 b_maxs.data_ptr(), b_seen.data_ptr(), st))
    nul = _lib.ArxSpan(None, None, 0, min(3, n), min(3, n))
    _lib.check(lib.arx_hash_minmax_float_consume(C.byref(nul), num_type, 1, 0.0, dgb.data_ptr(), min(3, n), b_mins.data_ptr(),
                                                 b_maxs.data_ptr(), b_seen.data_ptr(), st))
    _lib.check(lib.arx_hash_minmax_float_consume(C.byref(scal), num_type, 1, float("nan"), dgb.data_ptr(), k, b_mins.data_ptr(),
                                                 b_maxs.data_ptr(), b_seen.data_ptr(), st))
    _lib.check(lib.arx_hash_minmax_i64_merge(a_mins.data_ptr(), a_maxs.data_ptr(), a_seen.data_ptr(), b_mins.data_ptr(),
                                             b_maxs.data_ptr(), b_seen.data_ptr(), dperm.data_ptr(), nb, st))
    oa, ob = O.HashMinMaxState(dtype=dtype), O.HashMinMaxState(dtype=dtype)
    oa.resize(num_groups)
    ob.resize(nb)
    oa.consume(vals.values, vals.valid_bitmap(), vals.offset, gid_a)
    ob.consume(vals_b.values, vals_b.valid_bitmap(), vals_b.offset, gid_b)
    ob.consume(None, None, 0, gid_b[:k], scalar=(-7.5, True))
    ob.consume(None, None, 0, gid_b[:min(3, n)], scalar=(0.0, False))
    ob.consume(None, None, 0, gid_b[:k], scalar=(np.nan, True))
    oa.merge(ob, perm)
    tdt = torch.float64 if num_type == 9 else torch.float32
    for skip_nulls in (True, False):
        oa.skip_nulls = skip_nulls
        wmin, wmax, wvalid = oa.finalize()
        bits = torch.zeros((num_groups + 63) // 64, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        omin = torch.zeros(num_groups, dtype=tdt, device=dev)
        omax = torch.zeros(num_groups, dtype=tdt, device=dev)
        _lib.check(lib.arx_hash_minmax_float_finalize(a_mins.data_ptr(), a_maxs.data_ptr(), a_seen.data_ptr(), num_groups,
                                                      int(skip_nulls), num_type, omin.data_ptr(), omax.data_ptr(), bits.data_ptr(),
                                                      cnt.data_ptr(), st))
        got_valid = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[:num_groups].astype(bool)
        tag = f"hash_minmax_float[{np.dtype(dtype).name},n={n},G={num_groups},skip_nulls={skip_nulls}]"
        assert_equal(got_valid, wvalid, tag + " validity")
        assert int(cnt.item()) == int(wvalid.sum()), tag + " valid count"
        for got, want, what in ((omin.cpu().numpy(), wmin, "mins"), (omax.cpu().numpy(), wmax, "maxs")):
            g, w = got[wvalid], want[wvalid]
            assert_equal(np.isnan(g), np.isnan(w), tag + f" {what}: NaN groups")
            assert bool((g[~np.isnan(g)] == w[~np.isnan(w)]).all()), tag + f" {what}"     # (numeric: -0.0 == +0.0)
    if pa is not None and n > 0:      # the oracle itself against the reference
        o1 = O.HashMinMaxState(dtype=dtype)
        o1.resize(num_groups)
        o1.consume(vals.values, vals.valid_bitmap(), vals.offset, gid_a)
        t = pa.table({"g": pa.array(gid_a), "v": vals.to_pyarrow()})
        r = t.group_by("g", use_threads=False).aggregate([("v", "min"), ("v", "max")]).sort_by("g")
        gs = r.column("g").to_numpy()
        mn, mx, valid = o1.finalize()
        assert_equal(valid[gs], ~np.asarray(r.column("v_min").is_null()), "oracle float hash_min validity vs pyarrow")
        sel = valid[gs]
        for mine, col in ((mn, "v_min"), (mx, "v_max")):
            ref = r.column(col).drop_null().to_numpy()
            assert_equal(np.isnan(mine[gs][sel]), np.isnan(ref), f"oracle float {col}: NaN groups vs pyarrow")
            ok = ~np.isnan(ref)
            assert bool((mine[gs][sel][ok] == ref[ok]).all()), f"oracle float {col} vs pyarrow"


def check_sum_float(amd, rng, sizes, null_ps=(0.0, 0.01, 0.3, 0.97), dtypes=(np.float64, np.float32), oracle_max=300_000):
    """arx_sum_float: the float sum equals the reference's BIT FOR BIT (SumArray's pairwise summation is a fixed tree of
    additions) — against the oracle's restatement (sizes up to oracle_max) and against pyarrow's own pc.sum on the same
    host array; slices at odd offsets, all-null and all-valid bitmaps, NaN / inf / -0.0 among the values; and
    arx_reduce_float_minmax (order keys) against numpy's fmin / fmax."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    for dtype in dtypes:
        num_type = 9 if np.dtype(dtype) == np.float64 else 8
        for n in sizes:
            for null_p in null_ps:
                offset = int(rng.integers(0, 70)) if n else 0
                a = util.random_array(rng, dtype, n, null_p=null_p, offset=offset, tail=5)
                a.values *= (10.0 ** rng.integers(-15, 15, len(a.values))).astype(dtype)
                if n > 40:
                    a.values[offset + 7] = -0.0
                    if null_p == 0.01:
                        a.values[offset + 11] = np.inf
                    if null_p == 0.3 and n > 3000:
                        a.values[offset + 1000] = np.nan
                d = a.to_device(amd)
                sp = d.span()
                nulls = d.null_count if d.null_count is not None else -1
                ws_bytes = lib.arx_sum_float_workspace_bytes(n, nulls)
                ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
                out_sum, out_count = C.c_double(0), C.c_int64(0)
                _lib.check(lib.arx_sum_float(C.byref(sp), num_type, ws.data_ptr(), ws_bytes, C.byref(out_sum), C.byref(out_count), st))
                tag = f"sum_float[{np.dtype(dtype).name},n={n},null_p={null_p},offset={offset}]"
                vals = a.values[offset:offset + n]
                valid = None if a.valid is None else a.valid[offset:offset + n]
                want_count = n if valid is None else int(valid.sum())
                assert out_count.value == want_count, (tag, out_count.value, want_count)
                got = np.float64(out_sum.value)
                if n <= oracle_max:
                    want, _ = O.sum_float_pairwise(vals, valid)
                    assert got.tobytes() == np.float64(want).tobytes() or (np.isnan(got) and np.isnan(want)), (tag, "oracle", got, want)
                if pa is not None:
                    ref = pc.sum(pa.array(vals, mask=None if valid is None else ~valid), min_count=0).as_py()
                    assert got.tobytes() == np.float64(ref).tobytes() or (np.isnan(got) and np.isnan(ref)), (tag, "pyarrow", got, ref)
                # min / max by order keys
                acc = torch.zeros(4, dtype=torch.int64, device=dev)
                _lib.check(lib.arx_reduce_i64_init(acc.data_ptr(), st))
                _lib.check(lib.arx_reduce_float_minmax(C.byref(sp), num_type, acc.data_ptr(), st))
                h = acc.cpu().numpy()
                ok = np.ones(n, bool) if valid is None else valid
                assert int(h[1]) == want_count, (tag, "minmax count")
                live = vals[ok].astype(np.float64)
                live = live[~np.isnan(live)]
                if len(live):
                    def unkey(k):
                        k = np.int64(k)
                        u = np.uint64(k) if k >= 0 else ~(np.uint64(k) ^ np.uint64(1 << 63))
                        return np.array([u], np.uint64).view(np.float64)[0]
                    assert unkey(h[2]) == live.min() and unkey(h[3]) == live.max(), (tag, unkey(h[2]), live.min(), unkey(h[3]), live.max())
                else:
                    assert int(h[2]) == np.iinfo(np.int64).max and int(h[3]) == np.iinfo(np.int64).min, (tag, "anti-extrema")


def check_coalesce2(amd, rng, n=5000):
    """arx_coalesce2 (fill_null's kernel): every fixed width and boolean; fill = an array with nulls of its own, a
    valid scalar, a null scalar; values with and without a bitmap, at offsets — against numpy's where() and pyarrow's
    own coalesce on the same host arrays."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    for dtype in (np.bool_, np.int8, np.uint16, np.int32, np.float32, np.int64, np.float64):
        width = 0 if dtype == np.bool_ else np.dtype(dtype).itemsize
        for null_p in (0.0, 0.3, 1.0):
            for fill_kind in ("array", "scalar", "null_scalar"):
                a = util.random_array(rng, dtype, n, null_p=null_p, offset=int(rng.integers(0, 70)), tail=3)
                b = util.random_array(rng, dtype, n, null_p=0.2, offset=int(rng.integers(0, 70)), tail=3)
                da, db = a.to_device(amd), b.to_device(amd)
                sa, sb = da.span(), db.span()
                words = (n + 63) // 64
                out = torch.zeros(words * 8 + 8 if width == 0 else max(n * width, 8), dtype=torch.uint8, device=dev)
                ov = torch.zeros(words + 1, dtype=torch.int64, device=dev)
                sc_host = np.array([b.values[b.offset]], dtype=np.uint8 if width == 0 else dtype)
                fill_ptr = C.byref(sb) if fill_kind == "array" else None
                sc_ptr = sc_host.ctypes.data_as(C.c_void_p) if fill_kind == "scalar" else None
                _lib.check(lib.arx_coalesce2(width, C.byref(sa), fill_ptr, sc_ptr, n, out.data_ptr(), ov.data_ptr(), st))
                av = np.ones(n, bool) if a.valid is None else a.valid[a.offset:a.offset + n]
                avals = a.values[a.offset:a.offset + n]
                if fill_kind == "array":
                    bv = np.ones(n, bool) if b.valid is None else b.valid[b.offset:b.offset + n]
                    bvals = b.values[b.offset:b.offset + n]
                elif fill_kind == "scalar":
                    bv, bvals = np.ones(n, bool), np.full(n, b.values[b.offset], dtype)
                else:
                    bv, bvals = np.zeros(n, bool), np.zeros(n, dtype)
                want_valid = av | bv
                want = np.where(av, avals, bvals)
                got_valid = np.unpackbits(ov.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
                tag = f"coalesce2[{np.dtype(dtype).name},null_p={null_p},{fill_kind}]"
                assert_equal(got_valid, want_valid, tag + " validity")
                if width == 0:
                    got = np.unpackbits(out.cpu().numpy(), bitorder="little")[:n].astype(bool)
                else:
                    got = out.cpu().numpy()[: n * width].view(dtype)
                assert_equal(got[want_valid].view(np.uint8 if width == 0 else f"u{width}"),
                             want[want_valid].view(np.uint8 if width == 0 else f"u{width}"), tag + " values")
                if pa is not None:
                    pa_a = pa.array(avals, mask=~av)
                    pa_b = pa.array(bvals, mask=~bv) if fill_kind == "array" else pa.scalar(None if fill_kind == "null_scalar" else bvals[0].item(), pa_a.type)
                    ref = pc.coalesce(pa_a, pa_b)
                    assert_equal(np.asarray(ref.is_valid()), got_valid, tag + " validity vs pyarrow")
                    assert pa.array(got, mask=~got_valid).equals(ref), tag + " vs pyarrow"


def check_bitmap_copy_segments(amd, rng, scale=1):
    """arx_bitmap_copy_segments: bit ranges at any source bit offset ORed into zeroed bitmaps back to back (ranges meet
    inside words), NULL sources (= all ones), empty ranges, two destination bitmaps in one launch."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    lengths = [0, 1, 7, 63, 64, 65, 127, 128, 129, 4096, 32768, 100003 * scale, 5]
    src_bits_h = rng.integers(0, 2, 400_000 * scale + 64, dtype=np.uint8)
    src = to_device(np.packbits(src_bits_h, bitorder="little"), dev)
    total = sum(lengths) * 2
    dsts = [torch.zeros((total + 63) // 64 + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    want = [np.zeros(((total + 63) // 64 + 1) * 64, np.uint8) for _ in range(2)]
    pos = [3, 0]                     # the first bitmap does not even start at a word boundary
    segs = []
    for i, n in enumerate(lengths * 2):
        d = i % 2
        null_src = i % 5 == 3
        so = int(rng.integers(0, len(src_bits_h) - n - 1)) if n else int(rng.integers(0, 100))
        segs.append((0 if null_src else src.data_ptr(), so, dsts[d].data_ptr(), pos[d], n))
        want[d][pos[d]:pos[d] + n] = 1 if null_src else src_bits_h[so:so + n]
        pos[d] += n
    table = np.zeros(len(segs), BIT_SEG)
    for i, sg in enumerate(segs):
        table[i] = sg
    d_table = to_device(table.view(np.uint8), dev)
    _lib.check(lib.arx_bitmap_copy_segments(d_table.data_ptr(), len(segs), max(lengths), current_stream(dev)))
    for d in range(2):
        got = np.unpackbits(dsts[d].cpu().numpy().view(np.uint8), bitorder="little")
        assert_equal(got[: len(want[d])], want[d], f"bitmap_copy_segments dst {d}")


def _pack_bits(valid: np.ndarray, offset: int = 0) -> np.ndarray:
    """bool[n] -> LSB-first bitmap bytes with `offset` leading garbage-free bits, padded to whole 64-bit words"""
    bits = np.concatenate([np.zeros(offset, dtype=bool), valid])
    out = np.packbits(bits, bitorder="little")
    return np.concatenate([out, np.zeros((-len(out)) % 8 + 8, dtype=np.uint8)])


def check_take_rows(amd, rng, row_bytes_list=(1, 2, 3, 4, 12, 16, 20, 24, 48, 100, 512, 1000), n=3000, m=2500):
    """arx_take_rows (rows of any byte width: fixed_size_list of fixed-width values without nulls, FSLTakeExec ->
    FixedWidthTakeExec): every index type, null indices, null source rows at an offset, empty and single-row inputs; the
    output rows, validity words (padding bits clear) and the valid count against numpy."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    idx_types = [(np.uint8, 0), (np.int8, 1), (np.uint16, 2), (np.int16, 3), (np.uint32, 4), (np.int32, 5), (np.uint64, 6), (np.int64, 7)]
    case = 0
    for rb in row_bytes_list:
        for mm in (m, 1, 0, 64, 65):
            case += 1
            idt, tid = idx_types[case % len(idx_types)]
            voff = int(rng.integers(0, 9)) if case % 2 else 0
            nn = max(1, min(n, np.iinfo(idt).max))
            values = rng.integers(0, 256, (voff + nn) * rb + 32, dtype=np.uint8)
            has_sv, has_iv = case % 3 != 0, case % 4 != 1
            svalid = rng.random(nn) > 0.2 if has_sv else np.ones(nn, dtype=bool)
            ivalid = rng.random(mm) > 0.1 if has_iv else np.ones(mm, dtype=bool)
            idx = rng.integers(0, nn, mm).astype(idt)
            ioff = 3 if case % 5 == 0 else 0
            idx_buf = np.concatenate([np.zeros(ioff, dtype=idt), idx])
            d_values = to_device(values, dev)
            d_idx = to_device(idx_buf.view(np.uint8) if len(idx_buf) else np.zeros(8, dtype=np.uint8), dev)
            d_sv = to_device(_pack_bits(svalid, voff), dev) if has_sv else None
            d_iv = to_device(_pack_bits(ivalid, ioff), dev) if has_iv else None
            vs = _lib.ArxSpan(d_sv.data_ptr() if has_sv else None, d_values.data_ptr(), voff, nn, -1 if has_sv else 0)
            isp = _lib.ArxSpan(d_iv.data_ptr() if has_iv else None, d_idx.data_ptr(), ioff, mm, -1 if has_iv else 0)
            out = torch.full((max(mm * rb, 1) + 64,), 0xAB, dtype=torch.uint8, device=dev)
            ov = torch.full((((mm + 63) // 64) * 8 + 8,), 0xFF, dtype=torch.uint8, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            need_valid = has_sv or has_iv
            _lib.check(lib.arx_take_rows(C.byref(vs), rb, C.byref(isp), tid, out.data_ptr(), ov.data_ptr() if need_valid else None,
                                         cnt.data_ptr(), st))
            tag = f"take_rows[row_bytes={rb},m={mm},idx={np.dtype(idt).name},voff={voff},ioff={ioff},sv={has_sv},iv={has_iv}]"
            ok = ivalid & svalid[idx.astype(np.int64)] if mm else np.zeros(0, dtype=bool)
            rows = values[voff * rb: (voff + nn) * rb].reshape(nn, rb)
            want = np.where(ok[:, None], rows[idx.astype(np.int64)], 0).astype(np.uint8) if mm else np.zeros((0, rb), dtype=np.uint8)
            got = out.cpu().numpy()
            assert_equal(got[: mm * rb].reshape(mm, rb), want, tag + " rows")
            assert (got[mm * rb: mm * rb + 16] == 0xAB).all(), tag + " wrote past the output"
            assert int(cnt.item()) == int(ok.sum()), (tag, int(cnt.item()), int(ok.sum()))
            if need_valid and mm:
                words = ov.cpu().numpy()[: ((mm + 63) // 64) * 8]
                bits = np.unpackbits(words, bitorder="little")
                assert_equal(bits[:mm].astype(bool), ok, tag + " validity")
                assert not bits[mm:].any(), tag + " padding bits"
    # argument checks
    vs = _lib.ArxSpan(None, d_values.data_ptr(), 0, 1, 0)
    isp = _lib.ArxSpan(None, d_idx.data_ptr(), 0, 1, 0)
    assert lib.arx_take_rows(C.byref(vs), 0, C.byref(isp), 4, out.data_ptr(), None, None, st) == _lib.ARX_NOT_IMPLEMENTED
    assert lib.arx_take_rows(C.byref(vs), -4, C.byref(isp), 4, out.data_ptr(), None, None, st) == _lib.ARX_INVALID
    assert lib.arx_take_rows(C.byref(vs), 8, C.byref(isp), 9, out.data_ptr(), None, None, st) == _lib.ARX_NOT_IMPLEMENTED
    assert lib.arx_take_rows(None, 8, C.byref(isp), 4, out.data_ptr(), None, None, st) == _lib.ARX_INVALID


def check_list_take(amd, rng, n=2000, m=1700):
    """arx_binary_take_offsets + arx_(large_)list_take_data: a list whose nested values are fixed-width and free of nulls
    is a binary array whose offsets count elements (ListSelectionImpl, vector_selection_internal.cc:620-760) — element
    widths 1 .. 32 bytes, int32 and int64 offsets, sliced lists, null lists, null indices, empty lists."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    case = 0
    for shift in range(6):
        w = 1 << shift
        for large in (False, True):
            case += 1
            odt = np.int64 if large else np.int32
            voff = 5 if case % 2 else 0
            lens = rng.integers(0, 9, voff + n)
            lens[rng.random(voff + n) < 0.2] = 0
            offs = np.concatenate([[7], 7 + np.cumsum(lens)]).astype(odt)       # (a sliced child: offsets need not start at 0)
            child = rng.integers(0, 256, int(offs[-1]) * w + 64, dtype=np.uint8)
            svalid = rng.random(n) > 0.15
            ivalid = rng.random(m) > 0.1
            idx = rng.integers(0, n, m).astype(np.uint32)
            d_offs, d_child, d_idx = to_device(offs.view(np.uint8), dev), to_device(child, dev), to_device(idx.view(np.uint8), dev)
            d_sv, d_iv = to_device(_pack_bits(svalid, voff), dev), to_device(_pack_bits(ivalid, 0), dev)
            vs = _lib.ArxBinarySpan(d_sv.data_ptr(), d_offs.data_ptr(), d_child.data_ptr(), voff, n, -1)
            isp = _lib.ArxSpan(d_iv.data_ptr(), d_idx.data_ptr(), 0, m, -1)
            ws_bytes = (lib.arx_large_binary_take_workspace_bytes if large else lib.arx_binary_take_workspace_bytes)(m)
            ws = torch.zeros(ws_bytes + 64, dtype=torch.uint8, device=dev)
            out_offs = torch.zeros((m + 1) * offs.itemsize + 8, dtype=torch.uint8, device=dev)
            ov = torch.zeros(((m + 63) // 64) * 8 + 8, dtype=torch.uint8, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            total = C.c_int64(0)
            f_off = lib.arx_large_binary_take_offsets if large else lib.arx_binary_take_offsets
            _lib.check(f_off(C.byref(vs), C.byref(isp), 4, ws.data_ptr(), ws_bytes, out_offs.data_ptr(), ov.data_ptr(), cnt.data_ptr(),
                             C.byref(total), st))
            ok = ivalid & svalid[idx]
            src = idx.astype(np.int64) + voff
            want_lens = np.where(ok, lens[src], 0)
            want_offs = np.concatenate([[0], np.cumsum(want_lens)]).astype(odt)
            assert total.value == int(want_offs[-1]), (shift, large, total.value, int(want_offs[-1]))
            got_offs = out_offs.cpu().numpy()[: (m + 1) * offs.itemsize].view(odt)
            assert_equal(got_offs, want_offs, f"list_take offsets w={w} large={large}")
            data = torch.full((max(total.value * w, 1) + 32,), 0xCD, dtype=torch.uint8, device=dev)
            f_data = lib.arx_large_list_take_data if large else lib.arx_list_take_data
            _lib.check(f_data(C.byref(vs), shift, m, ws.data_ptr(), ws_bytes, out_offs.data_ptr(), total.value, data.data_ptr(), st))
            want = np.concatenate([child[int(offs[s]) * w: int(offs[s + 1]) * w] for s, o in zip(src, ok) if o] + [np.zeros(0, dtype=np.uint8)])
            got = data.cpu().numpy()
            assert_equal(got[: total.value * w], want, f"list_take data w={w} large={large}")
            assert (got[total.value * w: total.value * w + 16] == 0xCD).all()
            assert int(cnt.item()) == int(ok.sum())
    assert lib.arx_list_take_data(C.byref(vs), 6, m, ws.data_ptr(), ws_bytes, out_offs.data_ptr(), 1, data.data_ptr(), st) == _lib.ARX_INVALID


def check_hash_sum_float(amd, rng, n=20000, groups=(1, 7, 300, 5000), dtypes=(np.float64, np.float32)):
    """arx_hash_sum_float_consume / arx_hash_sum_f64_merge / arx_hash_mean_f64_finalize: the grouped float sum equals the
    reference's BIT FOR BIT — row-order double accumulation per group — over several batches (the state continues), values
    whose magnitudes differ by 30 orders (any other order of additions gives other bits), nulls at an offset, inf / NaN /
    -0.0, a broadcast scalar; the oracle's restatement is pinned to Table.group_by(use_threads=False) on the same rows."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    for dtype in dtypes:
        num_type = 9 if np.dtype(dtype) == np.float64 else 8
        for G in groups:
            sums = torch.zeros(G, dtype=torch.float64, device=dev)
            counts = torch.zeros(G, dtype=torch.int64, device=dev)
            seen = torch.zeros(G, dtype=torch.int32, device=dev)
            w_sums, w_counts, w_seen = np.zeros(G), np.zeros(G, dtype=np.int64), np.zeros(G, dtype=bool)
            all_v, all_valid, all_g = [], [], []
            for batch, nn in enumerate((n, 1, n // 3 + 5)):
                voff = int(rng.integers(0, 70))
                vals = (rng.standard_normal(voff + nn) * 10.0 ** rng.integers(-15, 15, voff + nn)).astype(dtype)
                if nn > 50:
                    vals[voff + 3], vals[voff + 11] = -0.0, np.float32(1e30) if dtype == np.float32 else 1e300
                    if batch == 2 and G > 1:
                        vals[voff + 17] = np.inf
                valid = rng.random(nn) > (0.1 if batch != 1 else 0.0)
                gids = rng.integers(0, G, nn).astype(np.uint32)
                if G > 5 and nn > 4000:
                    gids[rng.random(nn) < 0.6] = 3           # one long run (> 1024 rows: the wave-cooperative walker) among short ones
                d_vals, d_valid, d_gids = to_device(vals, dev), to_device(_pack_bits(valid, voff), dev), to_device(gids.view(np.uint8), dev)
                sp = _lib.ArxSpan(d_valid.data_ptr(), d_vals.data_ptr(), voff, nn, -1)
                ws_bytes = lib.arx_hash_sum_float_workspace_bytes(nn)
                ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)
                _lib.check(lib.arx_hash_sum_float_consume(C.byref(sp), num_type, 0, 0.0, d_gids.data_ptr(), nn, ws.data_ptr(), ws_bytes + 256,
                                                          sums.data_ptr(), counts.data_ptr(), seen.data_ptr(), st))
                O.hash_sum_float_row_order(vals[voff:], valid, gids, G, w_sums, w_counts, w_seen)
                all_v.append(vals[voff:]); all_valid.append(valid); all_g.append(gids)
                tag = f"hash_sum_float[{np.dtype(dtype).name},G={G},batch={batch}]"
                got = sums.cpu().numpy()
                same = (got.view(np.uint64) == w_sums.view(np.uint64)) | (np.isnan(got) & np.isnan(w_sums))
                assert same.all(), (tag, np.nonzero(~same)[0][:5], got[~same][:3], w_sums[~same][:3])
                assert_equal(counts.cpu().numpy(), w_counts, tag + " counts")
                assert_equal(seen.cpu().numpy().astype(bool), w_seen, tag + " null_seen")
            # a broadcast scalar: the same addend once per row, in row order
            gids = rng.integers(0, G, 200).astype(np.uint32)
            d_gids = to_device(gids.view(np.uint8), dev)
            sp = _lib.ArxSpan(None, None, 0, 200, 0)
            ws = torch.zeros(lib.arx_hash_sum_float_workspace_bytes(200) + 256, dtype=torch.uint8, device=dev)
            _lib.check(lib.arx_hash_sum_float_consume(C.byref(sp), num_type, 1, 0.1, d_gids.data_ptr(), 200, ws.data_ptr(), ws.numel(),
                                                      sums.data_ptr(), counts.data_ptr(), seen.data_ptr(), st))
            O.hash_sum_float_row_order(np.full(200, 0.1), None, gids, G, w_sums, w_counts, w_seen)
            got = sums.cpu().numpy()
            assert ((got.view(np.uint64) == w_sums.view(np.uint64)) | (np.isnan(got) & np.isnan(w_sums))).all(), "scalar addend"
            # mean = sum / count; merge into a second state through a mapping
            means = torch.zeros(G, dtype=torch.float64, device=dev)
            _lib.check(lib.arx_hash_mean_f64_finalize(sums.data_ptr(), counts.data_ptr(), G, means.data_ptr(), st))
            with np.errstate(all="ignore"):
                want_means = np.where(w_counts > 0, w_sums / np.maximum(w_counts, 1), 0.0)
            gm = means.cpu().numpy()
            assert ((gm.view(np.uint64) == want_means.view(np.uint64)) | (np.isnan(gm) & np.isnan(want_means))).all(), "means"
            perm = rng.permutation(G + 3)[:G].astype(np.uint32)
            t_sums = torch.full((G + 3,), 0.5, dtype=torch.float64, device=dev)
            t_counts = torch.ones(G + 3, dtype=torch.int64, device=dev)
            t_seen = torch.zeros(G + 3, dtype=torch.int32, device=dev)
            d_map = to_device(perm.view(np.uint8), dev)
            _lib.check(lib.arx_hash_sum_f64_merge(t_sums.data_ptr(), t_counts.data_ptr(), t_seen.data_ptr(), sums.data_ptr(), counts.data_ptr(),
                                                  seen.data_ptr(), d_map.data_ptr(), G, st))
            want_t = np.full(G + 3, 0.5)
            with np.errstate(all="ignore"):
                want_t[perm] = want_t[perm] + w_sums
            gt = t_sums.cpu().numpy()
            assert ((gt.view(np.uint64) == want_t.view(np.uint64)) | (np.isnan(gt) & np.isnan(want_t))).all(), "merge"
            assert t_counts.cpu().numpy()[perm].tolist() == (w_counts + 1).tolist()
            # the restatement itself against the reference: one thread, one batch = this row order
            if pa is not None and G > 1:
                v = np.concatenate(all_v)
                ok = np.concatenate(all_valid)
                g = np.concatenate(all_g)
                t = pa.table({"k": pa.array(g), "v": pa.array(v, mask=~ok)})
                ref = t.group_by("k", use_threads=False).aggregate([("v", "sum"), ("v", "mean")]).sort_by("k")
                o_sums, o_counts, _ = O.hash_sum_float_row_order(v, ok, g, G)
                keys = ref.column("k").to_numpy()
                rs = ref.column("v_sum").to_numpy(zero_copy_only=False).astype(np.float64)
                have = o_counts[keys] > 0
                a, b = rs[have], o_sums[keys][have]
                assert ((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all(), "oracle vs Table.group_by sum"
                rm = ref.column("v_mean").to_numpy(zero_copy_only=False).astype(np.float64)[have]
                with np.errstate(all="ignore"):
                    om = (o_sums[keys] / np.maximum(o_counts[keys], 1))[have]
                assert ((rm.view(np.uint64) == om.view(np.uint64)) | (np.isnan(rm) & np.isnan(om))).all(), "oracle vs Table.group_by mean"


def check_hash_sum_dec128(amd, rng, n=20000, groups=(1, 13, 4000)):
    """arx_hash_sum_dec128_consume / _merge / arx_dec128_pack: per-group sums of 128-bit two's-complement values modulo 2^128
    (BasicDecimal128 addition) — words that make the low half wrap on most additions, negative values, a hot group, nulls at an
    offset, several batches, a broadcast scalar — against Python integers; and Table.group_by's decimal128 sum on the same rows."""
    import decimal

    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    M = 1 << 128
    for G in groups:
        lo = torch.zeros(G, dtype=torch.int64, device=dev)
        hi = torch.zeros(G, dtype=torch.int64, device=dev)
        counts = torch.zeros(G, dtype=torch.int64, device=dev)
        seen = torch.zeros(G, dtype=torch.int32, device=dev)
        want = [0] * G
        w_counts, w_seen = np.zeros(G, dtype=np.int64), np.zeros(G, dtype=bool)
        rows = []
        for batch, nn in enumerate((n, 1, n // 2 + 3)):
            voff = int(rng.integers(0, 70))
            words = rng.integers(0, 2**64, (voff + nn, 2), dtype=np.uint64)          # [lo, hi]
            if batch == 0:
                words[voff:, 1] = np.where(rng.random(nn) < 0.5, 0, 2**64 - 1).astype(np.uint64)   # small magnitudes of both signs
            valid = rng.random(nn) > 0.1
            gids = rng.integers(0, G, nn).astype(np.uint32)
            if G > 5:
                gids[rng.random(nn) < 0.4] = 2                                     # a hot group
            d_vals, d_valid, d_gids = to_device(words.view(np.uint8).reshape(-1), dev), to_device(_pack_bits(valid, voff), dev), to_device(gids.view(np.uint8), dev)
            sp = _lib.ArxSpan(d_valid.data_ptr(), d_vals.data_ptr(), voff, nn, -1)
            _lib.check(lib.arx_hash_sum_dec128_consume(C.byref(sp), 0, 0, 0, d_gids.data_ptr(), nn, lo.data_ptr(), hi.data_ptr(), counts.data_ptr(),
                                                       seen.data_ptr(), st))
            for i in range(nn):
                g = int(gids[i])
                if valid[i]:
                    want[g] = (want[g] + int(words[voff + i, 0]) + (int(words[voff + i, 1]) << 64)) % M
                    w_counts[g] += 1
                else:
                    w_seen[g] = True
            rows.append((words[voff:], valid, gids))
            got = [(int(a) % 2**64) + ((int(b) % 2**64) << 64) for a, b in zip(lo.cpu().numpy().tolist(), hi.cpu().numpy().tolist())]
            assert got == want, (G, batch, [i for i in range(G) if got[i] != want[i]][:4])
            assert_equal(counts.cpu().numpy(), w_counts, f"dec128 counts G={G}")
            assert_equal(seen.cpu().numpy().astype(bool), w_seen, f"dec128 null_seen G={G}")
        # a broadcast scalar (-1: every addition of the low word wraps), 300 rows
        gids = rng.integers(0, G, 300).astype(np.uint32)
        d_gids = to_device(gids.view(np.uint8), dev)
        sp = _lib.ArxSpan(None, None, 0, 300, 0)
        _lib.check(lib.arx_hash_sum_dec128_consume(C.byref(sp), 1, 2**64 - 1, 2**64 - 1, d_gids.data_ptr(), 300, lo.data_ptr(), hi.data_ptr(),
                                                   counts.data_ptr(), seen.data_ptr(), st))
        for g in gids.tolist():
            want[g] = (want[g] - 1) % M
        got = [(int(a) % 2**64) + ((int(b) % 2**64) << 64) for a, b in zip(lo.cpu().numpy().tolist(), hi.cpu().numpy().tolist())]
        assert got == want, "scalar addend"
        # merge through a mapping, then pack
        perm = rng.permutation(G + 3)[:G].astype(np.uint32)
        t_lo = torch.full((G + 3,), -1, dtype=torch.int64, device=dev)
        t_hi = torch.full((G + 3,), 7, dtype=torch.int64, device=dev)
        t_counts = torch.ones(G + 3, dtype=torch.int64, device=dev)
        t_seen = torch.zeros(G + 3, dtype=torch.int32, device=dev)
        d_map = to_device(perm.view(np.uint8), dev)
        _lib.check(lib.arx_hash_sum_dec128_merge(t_lo.data_ptr(), t_hi.data_ptr(), t_counts.data_ptr(), t_seen.data_ptr(), lo.data_ptr(), hi.data_ptr(),
                                                 counts.data_ptr(), seen.data_ptr(), d_map.data_ptr(), G, st))
        base = (2**64 - 1) + (7 << 64)
        want_t = [base] * (G + 3)
        for g in range(G):
            want_t[int(perm[g])] = (base + want[g]) % M
        packed = torch.zeros((G + 3) * 16, dtype=torch.uint8, device=dev)
        _lib.check(lib.arx_dec128_pack(t_lo.data_ptr(), t_hi.data_ptr(), G + 3, packed.data_ptr(), st))
        pw = packed.cpu().numpy().view(np.uint64).reshape(-1, 2)
        assert [int(a) + (int(b) << 64) for a, b in pw.tolist()] == want_t, "merge + pack"
        # the reference on the same kind of rows (values small enough for precision 38)
        if pa is not None and G > 1:
            words, valid, gids = rows[0]
            ints = [int(w[0]) - (1 << 64) * (1 if w[1] else 0) for w in words.tolist()]          # hi is 0 or all ones in batch 0
            arr = pa.array([decimal.Decimal(x).scaleb(-3) if ok else None for x, ok in zip(ints, valid)], pa.decimal128(25, 3))
            ref = pa.table({"k": pa.array(gids), "v": arr}).group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k")
            sums = {}
            for x, ok, g in zip(ints, valid, gids.tolist()):
                if ok:
                    sums[g] = sums.get(g, 0) + x
            for k, v in zip(ref.column("k").to_pylist(), ref.column("v_sum").to_pylist()):
                assert (v is None and k not in sums) or int(v.scaleb(3)) == sums[k], (k, v)


def check_hash_minmax_dec128(amd, rng, n=20000, groups=(1, 13, 4000)):
    """arx_hash_minmax_dec128_consume / _finalize: per-group extrema of 128-bit two's-complement values in signed order —
    values of both signs whose halves disagree in order, a long run (the wave walker), all-null groups, nulls at an offset,
    several batches continuing the state — against Python integers."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C

    def signed(lo, hi):
        x = int(lo) + (int(hi) << 64)
        return x - (1 << 128) if x >= (1 << 127) else x

    for G in groups:
        mins = torch.full((G * 2,), 0x5A5A5A5A, dtype=torch.int64, device=dev)       # (any content: `seen` says what holds a value)
        maxs = torch.full((G * 2,), 0x5A5A5A5A, dtype=torch.int64, device=dev)
        seen = torch.zeros(G, dtype=torch.int32, device=dev)
        w_min, w_max, w_null = [None] * G, [None] * G, np.zeros(G, dtype=bool)
        for batch, nn in enumerate((n, 1, n // 2 + 3)):
            voff = int(rng.integers(0, 70))
            words = rng.integers(0, 2**64, (voff + nn, 2), dtype=np.uint64)          # [lo, hi]
            words[voff:, 1] = rng.choice(np.array([0, 1, 2**63 - 1, 2**63, 2**64 - 1, 2**64 - 2], dtype=np.uint64), nn)   # few high words: the low word decides
            valid = rng.random(nn) > 0.1
            gids = rng.integers(0, G, nn).astype(np.uint32)
            if G > 5:
                gids[rng.random(nn) < 0.4] = 2                                     # a long run
                valid[gids == 4] = False                                          # a group of nulls only
            d_vals, d_valid, d_gids = to_device(words.view(np.uint8).reshape(-1), dev), to_device(_pack_bits(valid, voff), dev), to_device(gids.view(np.uint8), dev)
            sp = _lib.ArxSpan(d_valid.data_ptr(), d_vals.data_ptr(), voff, nn, -1)
            ws_bytes = lib.arx_hash_minmax_dec128_workspace_bytes(nn)
            ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)
            _lib.check(lib.arx_hash_minmax_dec128_consume(C.byref(sp), d_gids.data_ptr(), nn, ws.data_ptr(), ws_bytes + 256, mins.data_ptr(),
                                                          maxs.data_ptr(), seen.data_ptr(), st))
            for i in range(nn):
                g = int(gids[i])
                if valid[i]:
                    x = signed(words[voff + i, 0], words[voff + i, 1])
                    w_min[g] = x if w_min[g] is None else min(w_min[g], x)
                    w_max[g] = x if w_max[g] is None else max(w_max[g], x)
                else:
                    w_null[g] = True
            gs, gmn, gmx = seen.cpu().numpy(), mins.cpu().numpy().view(np.uint64).reshape(-1, 2), maxs.cpu().numpy().view(np.uint64).reshape(-1, 2)
            for g in range(G):
                assert bool(gs[g] & 2) == (w_min[g] is not None) and bool(gs[g] & 1) == bool(w_null[g]), (G, batch, g, int(gs[g]))
                if w_min[g] is not None:
                    assert signed(*gmn[g]) == w_min[g] and signed(*gmx[g]) == w_max[g], (G, batch, g)
        for skip in (1, 0):
            bits = torch.full((((G + 63) // 64) * 8 + 8,), 0xFF, dtype=torch.uint8, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            _lib.check(lib.arx_hash_minmax_dec128_finalize(seen.data_ptr(), G, skip, bits.data_ptr(), cnt.data_ptr(), st))
            want = np.array([w_min[g] is not None and (skip == 1 or not w_null[g]) for g in range(G)])
            got = np.unpackbits(bits.cpu().numpy()[: ((G + 63) // 64) * 8], bitorder="little")
            assert_equal(got[:G].astype(bool), want, f"dec128 minmax validity G={G} skip={skip}")
            assert not got[G:].any() and int(cnt.item()) == int(want.sum())
    assert lib.arx_hash_minmax_dec128_workspace_bytes(0) == 0
    assert lib.arx_hash_minmax_dec128_consume(C.byref(sp), d_gids.data_ptr(), nn, ws.data_ptr(), 16, mins.data_ptr(), maxs.data_ptr(), seen.data_ptr(),
                                              st) == _lib.ARX_INVALID


def check_reduce_dec128(amd, rng, sizes=(0, 1, 63, 64, 65, 5000, 70001)):
    """arx_reduce_dec128: {sum modulo 2^128, count, min, max} of a decimal128 column — full-range words, nulls at an offset,
    all-null and empty inputs — against Python integers."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    ws_bytes = lib.arx_reduce_dec128_workspace_bytes()
    ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)

    def signed(x):
        return x - (1 << 128) if x >= (1 << 127) else x

    for n in sizes:
        for null_p in (0.0, 0.2, 1.0):
            voff = int(rng.integers(0, 70)) if n else 0
            words = rng.integers(0, 2**64, (voff + max(n, 1), 2), dtype=np.uint64)
            valid = rng.random(n) >= null_p if null_p < 1.0 else np.zeros(n, dtype=bool)
            d_vals = to_device(words.view(np.uint8).reshape(-1), dev)
            d_valid = to_device(_pack_bits(valid, voff), dev)
            sp = _lib.ArxSpan(d_valid.data_ptr() if null_p else None, d_vals.data_ptr(), voff, n, -1 if null_p else 0)
            out = (C.c_uint64 * 8)()
            _lib.check(lib.arx_reduce_dec128(C.byref(sp), ws.data_ptr(), ws_bytes + 256, out, st))
            vals = [int(words[voff + i, 0]) + (int(words[voff + i, 1]) << 64) for i in range(n) if valid[i]]
            tag = f"reduce_dec128[n={n},null_p={null_p}]"
            assert out[2] == len(vals) and out[3] == (1 if vals else 0), (tag, out[2], out[3])
            assert out[0] + (out[1] << 64) == sum(vals) % (1 << 128), tag
            if vals:
                sv = [signed(v) for v in vals]
                assert signed(out[4] + (out[5] << 64)) == min(sv) and signed(out[6] + (out[7] << 64)) == max(sv), tag
    assert lib.arx_reduce_dec128(C.byref(sp), ws.data_ptr(), 16, out, st) == _lib.ARX_INVALID


def check_hash_product_and_edge_rows(amd, rng, n=20000, groups=(1, 7, 300, 5000)):
    """arx_hash_product_init / _consume (hash_product: wrapping integer products of every width, double products in ROW order —
    bit for bit, several batches, one long run for the wave-cooperative walker, nulls at an offset, zeros / inf / NaN / -0.0)
    and arx_group_edge_rows (the row of every group's first / last non-null value + the bitmap of groups that have one:
    hash_first / hash_last / hash_one) and arx_dec128_split, each against the oracle's restatement of the reference."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    num_types = {np.int8: 0, np.uint8: 1, np.int16: 2, np.uint16: 3, np.int32: 4, np.uint32: 5, np.int64: 6, np.uint64: 7, np.float32: 8, np.float64: 9}
    for dtype in (np.int64, np.uint64, np.int8, np.uint16, np.int32, np.float64, np.float32):
        is_float = np.dtype(dtype).kind == "f"
        for G in groups:
            prods = torch.zeros(G, dtype=torch.int64, device=dev)
            counts = torch.zeros(G, dtype=torch.int64, device=dev)
            seen = torch.zeros(G, dtype=torch.int32, device=dev)
            _lib.check(lib.arx_hash_product_init(prods.data_ptr(), num_types[dtype], G, st))
            w_p = w_c = w_s = None
            for batch, nn in enumerate((n, 1, n // 3 + 5)):
                voff = int(rng.integers(0, 70))
                if is_float:
                    vals = (1.0 + rng.standard_normal(voff + nn) * 0.5).astype(dtype)
                    if nn > 50:
                        vals[voff + 3], vals[voff + 11] = -0.0, dtype(1e30)
                        if batch == 2 and G > 1:
                            vals[voff + 17], vals[voff + 19] = np.inf, np.nan
                else:
                    info = np.iinfo(dtype)
                    vals = rng.integers(max(info.min, -7), min(info.max, 9), voff + nn).astype(dtype)
                    vals[rng.random(voff + nn) < 0.3] = dtype(info.max - 2) if info.max > 300 else dtype(3)     # (products that wrap many times)
                valid = rng.random(nn) > (0.1 if batch != 1 else 0.0)
                gids = rng.integers(0, G, nn).astype(np.uint32)
                if G > 5 and nn > 4000:
                    gids[rng.random(nn) < 0.6] = 3
                d_vals, d_valid, d_gids = to_device(vals, dev), to_device(_pack_bits(valid, voff), dev), to_device(gids.view(np.uint8), dev)
                sp = _lib.ArxSpan(d_valid.data_ptr(), d_vals.data_ptr(), voff, nn, -1)
                ws_bytes = lib.arx_hash_sum_float_workspace_bytes(nn)
                ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)
                _lib.check(lib.arx_hash_product_consume(C.byref(sp), num_types[dtype], d_gids.data_ptr(), nn, ws.data_ptr(), ws_bytes + 256,
                                                        prods.data_ptr(), counts.data_ptr(), seen.data_ptr(), st))
                w_p, w_c, w_s = O.hash_product_row_order(vals[voff:], valid, gids, G, w_p, w_c, w_s)
                tag = f"hash_product[{np.dtype(dtype).name},G={G},batch={batch}]"
                got = prods.cpu().numpy().view(np.uint64)
                want = w_p.view(np.uint64)
                same = (got == want) | (is_float & np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64)))
                assert same.all(), (tag, np.nonzero(~same)[0][:5], got[~same][:3], want[~same][:3])
                assert_equal(counts.cpu().numpy(), w_c, tag + " counts")
                assert_equal(seen.cpu().numpy().astype(bool), w_s, tag + " null_seen")
    # ---- arx_group_edge_rows
    for G in groups:
        for nn, null_p, voff in ((n, 0.3, 5), (n // 7 + 1, 0.0, 0), (200, 1.0, 3), (1, 0.0, 0)):
            valid = rng.random(nn) >= null_p
            gids = rng.integers(0, G, nn).astype(np.uint32)
            d_gids = to_device(gids.view(np.uint8), dev)
            d_valid = to_device(_pack_bits(valid, voff), dev)
            for last in (0, 1):
                for use_bits in (True, False):
                    rows = torch.full((G,), 7, dtype=torch.int32, device=dev)
                    has = torch.zeros((G + 63) // 64 * 8 + 8, dtype=torch.uint8, device=dev)
                    _lib.check(lib.arx_group_edge_rows(d_gids.data_ptr(), d_valid.data_ptr() if use_bits else None, voff if use_bits else 0, nn, G, last,
                                                       rows.data_ptr(), has.data_ptr(), st))
                    w_rows, w_has = O.group_edge_rows(gids, valid if use_bits else None, G, last=bool(last))
                    tag = f"group_edge_rows[G={G},n={nn},last={last},bits={use_bits}]"
                    got_has = np.unpackbits(has.cpu().numpy(), bitorder="little")[:G].astype(bool)
                    assert_equal(got_has, w_has, tag + " has_row")
                    assert_equal(rows.cpu().numpy().view(np.uint32), w_rows, tag + " rows")
                    assert not np.unpackbits(has.cpu().numpy(), bitorder="little")[G:(G + 63) // 64 * 64].any(), tag + " padding bits"
    # ---- arx_dec128_split
    for nn in (0, 1, 1000, n):
        words = rng.integers(0, 1 << 64, size=2 * max(nn, 1), dtype=np.uint64)
        d_in = to_device(words.view(np.uint8), dev)
        lo = torch.zeros(max(nn, 1), dtype=torch.int64, device=dev)
        hi = torch.zeros(max(nn, 1), dtype=torch.int64, device=dev)
        _lib.check(lib.arx_dec128_split(d_in.data_ptr(), nn, lo.data_ptr(), hi.data_ptr(), st))
        assert_equal(lo.cpu().numpy().view(np.uint64)[:nn], words[0:2 * nn:2], "dec128_split lo")
        assert_equal(hi.cpu().numpy().view(np.uint64)[:nn], words[1:2 * nn:2], "dec128_split hi")



def check_group_moments(amd, rng, n=20000, groups=(1, 7, 300, 5000)):
    """arx_group_central_power + arx_hash_moments_finalize (hash_variance / hash_stddev / hash_skew / hash_kurtosis): pass 1 =
    the row-order float sum, pass 2 = the row-order sums of (x - mean)^k — over ONE batch that is the reference's
    ConsumeGeneric operation for operation, so m2 and the variance / stddev are compared bit for bit with the oracle's
    restatement (pinned on the reference build, tests/test_oracle_pin.py) and skew / kurtosis to 1e-12 (their last
    subtraction may be fused on either side); nulls at an offset, a group of one value, of equal values (0 / 0 = NaN)."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    st = current_stream(dev)
    C = _lib.C
    for dtype, num_type in ((np.float64, 9), (np.float32, 8)):
        for G in groups:
            voff = int(rng.integers(0, 70))
            vals = (1e3 + rng.standard_normal(voff + n) * 7).astype(dtype)
            valid = rng.random(n) > 0.15
            gids = rng.integers(0, G, n).astype(np.uint32)
            if G > 5:
                gids[rng.random(n) < 0.5] = 3                         # one long group (the wave-cooperative walker)
                vals[voff:][gids == 4] = dtype(2.5)                   # a constant group: m2 = 0, skew / kurtosis 0 / 0
                one = np.nonzero(gids == 5)[0]
                valid[one[1:]] = False                                # a group of one value (and nulls)
            d_vals, d_valid, d_gids = to_device(vals, dev), to_device(_pack_bits(valid, voff), dev), to_device(gids.view(np.uint8), dev)
            sp = _lib.ArxSpan(d_valid.data_ptr(), d_vals.data_ptr(), voff, n, -1)
            ws_bytes = lib.arx_hash_sum_float_workspace_bytes(n)
            ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)
            z = lambda dt: torch.zeros(G, dtype=dt, device=dev)  # noqa: E731
            sums, counts, seen = z(torch.float64), z(torch.int64), z(torch.int32)
            _lib.check(lib.arx_hash_sum_float_consume(C.byref(sp), num_type, 0, 0.0, d_gids.data_ptr(), n, ws.data_ptr(), ws_bytes + 256,
                                                      sums.data_ptr(), counts.data_ptr(), seen.data_ptr(), st))
            dev_col = torch.empty(voff + n, dtype=torch.float64, device=dev)
            m = {}
            for power in (2, 3, 4):
                _lib.check(lib.arx_group_central_power(C.byref(sp), num_type, d_gids.data_ptr(), n, sums.data_ptr(), counts.data_ptr(), power,
                                                       dev_col.data_ptr() + 8 * voff, st))
                dsp = _lib.ArxSpan(d_valid.data_ptr(), dev_col.data_ptr(), voff, n, -1)
                m[power], c2, s2 = z(torch.float64), z(torch.int64), z(torch.int32)
                _lib.check(lib.arx_hash_sum_float_consume(C.byref(dsp), 9, 0, 0.0, d_gids.data_ptr(), n, ws.data_ptr(), ws_bytes + 256,
                                                          m[power].data_ptr(), c2.data_ptr(), s2.data_ptr(), st))
                assert_equal(c2.cpu().numpy(), counts.cpu().numpy(), "moment counts")
            want, _ = O.grouped_moments(vals[voff:], valid, gids, G, 4)
            tag = f"group_moments[{np.dtype(dtype).name},G={G}]"
            assert_equal(counts.cpu().numpy(), np.array([w[0] for w in want], dtype=np.int64), tag + " counts")
            for power, idx in ((2, 2), (3, 3), (4, 4)):
                got = m[power].cpu().numpy()
                exp = np.array([w[idx] for w in want], dtype=np.float64)
                assert (got.view(np.uint64) == exp.view(np.uint64)).all(), (tag, power, got[:4], exp[:4])
            for stat, ddof, biased in ((0, 0, 1), (0, 1, 1), (1, 2, 1), (2, 0, 1), (2, 0, 0), (3, 0, 1), (3, 0, 0)):
                out = torch.full((G,), 7.0, dtype=torch.float64, device=dev)
                _lib.check(lib.arx_hash_moments_finalize(counts.data_ptr(), m[2].data_ptr(), m[3].data_ptr(), m[4].data_ptr(), G, stat, ddof,
                                                         biased, out.data_ptr(), st))
                got = out.cpu().numpy()
                for g in range(G):
                    w = O.moments_statistic(want[g], stat, ddof, bool(biased))
                    if w is None:
                        assert got[g] == 0.0, (tag, stat, g, got[g])
                    elif np.isnan(w):
                        assert np.isnan(got[g]), (tag, stat, g, got[g])
                    elif stat <= 1:
                        assert np.float64(got[g]).view(np.uint64) == np.float64(w).view(np.uint64), (tag, stat, ddof, g, got[g], w)
                    else:
                        assert abs(got[g] - w) <= 1e-12 * max(1.0, abs(w)), (tag, stat, biased, g, got[g], w)
    # bad arguments are refused
    assert lib.arx_group_central_power(None, 9, None, 5, None, None, 2, None, st) == _lib.ARX_INVALID
    assert lib.arx_hash_moments_finalize(None, None, None, None, 3, 9, 0, 1, None, st) == _lib.ARX_INVALID

# --------------------------------------------------------------------------- rank / select_k / partition_nth
def rank_golden_cases():
    """(values, valid, order, null_placement, tiebreaker, expected) of tests/golden/rank_vectors.json — the known answers
    of the reference's TestRank (vector_sort_test.cc:2408-2507)."""
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rank_vectors.json")) as f:
        g = json.load(f)
    def arr(block, dtype):
        v = np.array([np.nan if x == "NaN" else x for x in block["values"]], dtype=dtype)
        return v, (None if block["valid"] is None else np.array(block["valid"], bool))
    out = []
    for dtype in (np.float64, np.float32):
        v, valid = arr(g["simple"], dtype)
        for place in ("at_end", "at_start"):
            for tb in ("min", "max", "first", "dense"):
                out.append((v, valid, "ascending", place, tb, g["simple"]["ascending"]))
                out.append((v, valid, "descending", place, tb, g["simple"]["descending"]))
        for name in ("all_tiebreakers", "nans_and_nulls"):
            v, valid = arr(g[name], dtype)
            out += [(v, valid, o, p, t, e) for o, p, t, e in g[name]["expected"]]
    # TestRank.Integral (:2509-2525) uses the same expected ranks on [2, 3, 1, 0, 5] and [1, 0, 5, null, 5, null, 0]
    for dtype in (np.int64, np.uint64, np.int32, np.uint32):
        v = np.array([2, 3, 1, 0, 5], dtype)
        for place in ("at_end", "at_start"):
            for tb in ("min", "max", "first", "dense"):
                out.append((v, None, "ascending", place, tb, g["simple"]["ascending"]))
                out.append((v, None, "descending", place, tb, g["simple"]["descending"]))
        v = np.array([1, 0, 5, 0, 5, 0, 0], dtype)
        valid = np.array(g["all_tiebreakers"]["valid"], bool)
        out += [(v, valid, o, p, t, e) for o, p, t, e in g["all_tiebreakers"]["expected"]]
    return out


# rank_normal: the reference's own bar for NormalPPF is EXPECT_DOUBLE_EQ = 4 ULPs (util/math_test.cc:28-82; its rank_normal
# test accepts 1e-8 absolute, vector_sort_test.cc:2656-2660).  The centre of the distribution is +, *, / only and comes out
# bit for bit; the tails go through log(), where the device's and the host's libraries may round differently.
RANK_NORMAL_ULPS = 4


def max_ulps(a, b) -> int:
    """Largest distance in units of the last place between two float64 arrays (equal infinities and equal zeros: 0)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0
    assert not (np.isnan(a).any() or np.isnan(b).any())
    ia, ib = a.view(np.int64).copy(), b.view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]          # two's-complement order of the doubles
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    return int(np.abs(ia.astype(object) - ib.astype(object)).max())


def check_rank(amd, rng_for, scale=1, light=False):
    """rank / rank_quantile on the device (arx_sort_indices + arx_rank) against the oracle's restatement of
    vector_rank.cc — bit-exact, quantile ranks included — on the reference's own known answers and on seeded arrays:
    all six key types, every tiebreaker, both orders and null placements, nulls, NaNs, signed zeros, long runs of ties
    that span tiles, sliced inputs, lengths around the tile size, empty."""
    for v, valid, order, place, tb, expected in rank_golden_cases():
        if light and v.dtype not in (np.float64, np.int32):      # (the emulated tier: two of the six types)
            continue
        d = util.HostArray(v, valid, 0, len(v)).to_device(amd)
        got = _data_np(amd.compute.rank(d, order, place, tb), np.uint64)
        assert_equal(got, np.array(expected, np.uint64), f"rank golden [{v.dtype},{order},{place},{tb}]")
    sizes = [0, 1, 2049] if light else [0, 1, 2047, 2048, 2049, 5000 * scale, 40_000 * scale]
    combos = [(o, p_) for o in ("ascending", "descending") for p_ in ("at_end", "at_start")]
    for dtype in (np.uint64, np.int64, np.uint32, np.int32, np.float64, np.float32):
        for n in sizes:
            rng = rng_for("rank", np.dtype(dtype).name, n)
            # (light — the emulated tier: one order / placement per array, every tiebreaker; all four on the GPU)
            chosen = [combos[int(rng.integers(0, 4))]] if light else combos
            kind = np.dtype(dtype).kind
            distinct = int(rng.choice([1, 3, 50, 1 << 30]))         # (1: one run over every tile; 3: runs of thousands)
            if kind == "f":
                vals = rng.integers(-distinct, distinct, n + 5, endpoint=True).astype(dtype) * dtype(0.5)
                vals[rng.random(n + 5) < 0.1] = np.nan
                vals[rng.random(n + 5) < 0.05] = dtype(-0.0)
            else:
                lo = 0 if kind == "u" else -distinct
                vals = rng.integers(lo, distinct, n + 5, endpoint=True).astype(dtype)
            null_p = float(rng.choice([0.0, 0.2]))
            valid = rng.random(n + 5) >= null_p if null_p else None
            arr = util.HostArray(vals, valid, 3, n)
            d = arr.to_device(amd)
            lv = arr.logical_values()
            lvalid = None if valid is None else arr.logical_valid()
            for order, place in chosen:
                if True:
                    for tb in ("min", "max", "first", "dense"):
                        got = amd.compute.rank(d, order, place, tb)
                        assert got.type == amd.array.uint64 and got.null_count == 0 and got.length == n
                        want = O.rank(lv, lvalid, order == "descending", place == "at_start", tb)
                        assert_equal(_data_np(got, np.uint64), want, f"rank[{np.dtype(dtype).name},n={n},{order},{place},{tb}]")
                    gq = amd.compute.rank_quantile(d, order, place)
                    wq = O.rank(lv, lvalid, order == "descending", place == "at_start", "quantile")
                    assert gq.type == amd.array.float64
                    assert_equal(_data_np(gq, np.float64).view(np.uint64), wq.view(np.uint64), f"rank_quantile[{np.dtype(dtype).name},n={n},{order},{place}]")
                    gn = amd.compute.rank_normal(d, order, place)
                    wn = O.rank(lv, lvalid, order == "descending", place == "at_start", "normal")
                    assert gn.type == amd.array.float64 and gn.null_count == 0
                    assert max_ulps(_data_np(gn, np.float64), wn) <= RANK_NORMAL_ULPS, f"rank_normal[{np.dtype(dtype).name},n={n},{order},{place}]"
    if pc is not None:     # the restatement itself against the reference build, on the way
        a = pa.array([1.5, None, float("nan"), -0.0, 0.0, 1.5])
        v = np.array([1.5, 0, np.nan, -0.0, 0.0, 1.5])
        valid = np.array([True, False, True, True, True, True])
        for tb in ("min", "max", "first", "dense"):
            assert_equal(O.rank(v, valid, False, False, tb), pc.rank(a, sort_keys="ascending", tiebreaker=tb).to_numpy(), "oracle vs pyarrow")


def check_sort_boolean_keys(amd, rng_for, scale=1):
    """array_sort_indices of boolean arrays (the reference's counting sort) — exact indices against pyarrow and the
    oracle's multi-key restatement: nulls, slices with bit offsets, every order / placement, all-null, empty."""
    for n in (0, 1, 63, 64, 65, 5000 * scale):
        rng = rng_for("sort-bool", n)
        for null_p in (0.0, 0.2, 1.0):
            arr = util.random_array(rng, np.bool_, n, null_p=null_p, offset=int(rng.integers(0, 70)), tail=3)
            d = arr.to_device(amd)
            for order in ("ascending", "descending"):
                for place in ("at_end", "at_start"):
                    got = _data_np(amd.compute.sort_indices(d, order, place), np.uint64)
                    want = O.sort_indices_multi([(arr.logical_values().astype(np.uint8), arr.logical_valid() if arr.valid is not None else None)],
                                                [order == "descending"], place == "at_start")
                    assert_equal(got, want, f"sort_indices[bool,n={n},nulls={null_p},{order},{place}]")
                    if pc is not None:
                        ref = pc.array_sort_indices(arr.to_pyarrow(), order=order, null_placement=place)
                        assert_equal(got, ref.to_numpy(), "boolean sort vs pyarrow")


def check_select_k_partition_nth(amd, rng_for, scale=1, light=False):
    """select_k_unstable and partition_nth_indices on the device.  Both promise a property, not one permutation
    (std::nth_element / heaps in the reference, vector_array_sort.cc:56-95, vector_select_k.cc:103-232): checked are
    the VALUES at the returned indices against the oracle's sorted order (select_k) and the partition property
    itself (partition_nth): a permutation, the pivot first ones no greater than the rest, null-likes at their end."""
    for dtype in ((np.int64, np.float32) if light else (np.int64, np.uint32, np.float64, np.float32)):
        for n in ((0, 1, 3000) if light else (0, 1, 777, 30_000 * scale)):
            rng = rng_for("selectk", np.dtype(dtype).name, n)
            if np.dtype(dtype).kind == "f":
                vals = rng.integers(-40, 40, n + 2).astype(dtype)
                vals[rng.random(n + 2) < 0.1] = np.nan
            else:
                vals = rng.integers(0, 80, n + 2).astype(dtype)
            valid = rng.random(n + 2) >= 0.15
            arr = util.HostArray(vals, valid, 1, n)
            d = arr.to_device(amd)
            lv, lvalid = arr.logical_values(), arr.logical_valid()
            null_like = ~lvalid | (np.isnan(lv) if lv.dtype.kind == "f" else np.zeros(n, bool))
            for place in ("at_end", "at_start"):
                for order in ("ascending", "descending"):
                    full = O.sort_indices(np.ascontiguousarray(arr.values), arr.valid_bitmap(), arr.offset, n,
                                          descending=(order == "descending"), nulls_at_start=(place == "at_start")).astype(np.int64)
                    for k in sorted({n // 3, n + 5} if light else {0, 1, n // 3, n, n + 5}):
                        got = _data_np(amd.compute.select_k_unstable(d, k, order, place), np.uint64).astype(np.int64)
                        want = full[:k]
                        assert len(got) == min(k, n)
                        # same values (and same null / NaN classes) position by position; indices may differ inside ties
                        assert_equal(lvalid[got], lvalid[want], "select_k validity")
                        both = lvalid[got]
                        gv, wv = lv[got][both], lv[want][both]
                        assert_equal(np.isnan(gv) if gv.dtype.kind == "f" else gv, np.isnan(wv) if wv.dtype.kind == "f" else wv, "select_k NaNs")
                        ok = ~np.isnan(gv) if gv.dtype.kind == "f" else np.ones(len(gv), bool)
                        assert_equal(gv[ok], wv[ok], f"select_k values [{np.dtype(dtype).name},n={n},k={k},{order},{place}]")
                for pivot in sorted({n // 2, n} if light else {0, n // 2, max(n - 1, 0), n}):
                    got = _data_np(amd.compute.partition_nth_indices(d, pivot, place), np.uint64).astype(np.int64)
                    assert_equal(np.sort(got), np.arange(n), "partition_nth: a permutation")
                    nl = null_like[got]
                    cnt = int(null_like.sum())
                    if place == "at_end":
                        assert not nl[:n - cnt].any() and nl[n - cnt:].all()
                        body, at = got[:n - cnt], pivot
                    else:
                        assert nl[:cnt].all() and not nl[cnt:].any()
                        body, at = got[cnt:], pivot - cnt
                    if 0 <= at < len(body):
                        assert (lv[body[:at]] <= lv[body[at]]).all() and (lv[body[at:]] >= lv[body[at]]).all(), "partition property"
            with pytest.raises(amd.ArrowIndexError):
                amd.compute.partition_nth_indices(d, n + 1)
    # select_k by a THRESHOLD (no sort of the column): forced at these sizes; the result is index for index the head of the
    # stable sort — full-range keys, keys from a narrow window (shared top bits), many ties, nulls at the end
    saved = amd.compute.SELECT_K_MIN_ROWS
    amd.compute.SELECT_K_MIN_ROWS = 0
    try:
        for dtype in (np.int64, np.uint64):
            for lo, hi in ((None, None), (1_700_000_000_000, 1_700_000_900_000), (0, 50), (0, 2)):
                n = (20_000 if light else 60_000 * scale)
                rng = rng_for("selectk-threshold", np.dtype(dtype).name, lo)
                arr = util.random_array(rng, dtype, n, null_p=0.1, offset=5, lo=lo, hi=hi)
                d = arr.to_device(amd)
                for order in ("ascending", "descending"):
                    full = O.sort_indices(np.ascontiguousarray(arr.values), arr.valid_bitmap(), arr.offset, n,
                                          descending=(order == "descending"), nulls_at_start=False)
                    for k in (1, 37, n // 20):
                        before = dict(amd.compute._SELECT_COUNTERS)
                        got = _data_np(amd.compute.select_k_unstable(d, k, order, "at_end"), np.uint64)
                        assert_equal(got, full[:k], f"select_k by threshold [{np.dtype(dtype).name},{lo},{order},k={k}]")
                        took = {name: amd.compute._SELECT_COUNTERS[name] - before[name] for name in before}
                        # (3 distinct keys: one bin holds more than a quarter of the rows — the sort is the better plan and is taken)
                        assert took == ({"threshold": 0, "sorted": 1} if hi == 2 else {"threshold": 1, "sorted": 0}), (took, lo, hi, k)
    finally:
        amd.compute.SELECT_K_MIN_ROWS = saved
