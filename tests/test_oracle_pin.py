"""Pins the oracle (oracle/arx_oracle.c) before anything trusts it:
 1. against golden vectors transcribed from the reference's unit tests
    (tests/golden/reference_vectors.json, each with its file:line);
 2. against the reference's own build — pyarrow 25.0.0 (libarrow.so.2500) — on seeded random
    grids shaped like the reference's randomized tests (FilterRandomTest, TakeRandomTest, ...);
 3. against fixtures generated from pyarrow by tests/golden/make_golden.py (so the pin survives
    on a machine without the wheel).
CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

from . import util as U
from .util import pa, pc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))


def from_list(xs, dtype):
    valid = np.array([x is not None for x in xs], dtype=bool)
    fill = False if dtype == np.bool_ else 0
    vals = np.array([fill if x is None else x for x in xs], dtype=dtype)
    return U.HostArray(vals, None if valid.all() else valid, 0, len(xs))


def to_list(vals, bm, n):
    valid = O.unpack_bits(bm, 0, n) if bm is not None else np.ones(n, bool)
    return [v if ok else None for v, ok in zip(vals.tolist(), valid.tolist())]


def o_filter(v: U.HostArray, m: U.HostArray, sel):
    code = 1 if sel == "emit_null" else 0
    vals, bm = O.filter(v.data_bytes(), v.valid_bitmap(), v.offset, m.data_bytes(), m.valid_bitmap(),
                        m.offset, m.length, code, True)
    return to_list(vals, bm, len(vals))


def golden_sort_array(values, dtype):
    """JSON list (None = null, "NaN") -> HostArray of `dtype`; None if the values do not fit the type."""
    if any(isinstance(x, float) and x != int(x) for x in values if x is not None and x != "NaN") and np.dtype(dtype).kind != "f":
        return None
    if any(x == "NaN" for x in values) and np.dtype(dtype).kind != "f":
        return None
    vals = [np.nan if x == "NaN" else x for x in values]
    return from_list(vals, dtype)


# ------------------------------------------------------------------ 1. golden vectors
@pytest.mark.parametrize("case", GOLD["get_take_indices"], ids=lambda c: c["cite"])
def test_golden_get_take_indices(case):
    m = from_list(case["mask"], np.bool_)
    code = 1 if case["sel"] == "emit_null" else 0
    vals, bm = O.mask_to_indices(m.data_bytes(), m.valid_bitmap(), 0, m.length, code, True)
    assert vals.dtype == np.uint16
    assert to_list(vals, bm, len(vals)) == case["want"]


@pytest.mark.parametrize("case", GOLD["filter_emit_null"], ids=lambda c: c["cite"])
@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64, np.float64])
def test_golden_filter(case, dtype):
    v, m = from_list(case["values"], dtype), from_list(case["mask"], np.bool_)
    assert o_filter(v, m, "emit_null") == case["want"]
    # AssertFilter also checks DROP == the expectation with mask-null rows removed (:258-284)
    drop_want = []
    it = iter(case["want"])
    for x in case["mask"]:
        if x is None:
            next(it)
        elif x:
            drop_want.append(next(it))
    assert o_filter(v, m, "drop") == drop_want


def test_golden_filter_sliced_mask():
    c = GOLD["filter_sliced_mask"]
    v = from_list(c["values"], np.int64)
    full = from_list(c["mask_full"], np.bool_)
    m = U.HostArray(full.values, full.valid, c["mask_offset"], c["mask_length"])
    assert o_filter(v, m, "drop") == c["want"] == o_filter(v, m, "emit_null")


@pytest.mark.parametrize("case", GOLD["take"], ids=lambda c: c["cite"])
@pytest.mark.parametrize("idx_dtype", [np.int8, np.uint32, np.int64])
def test_golden_take(case, idx_dtype):
    v, i = from_list(case["values"], np.int8), from_list(case["indices"], idx_dtype)
    vals, bm, vc = O.take(v.data_bytes(), v.valid_bitmap(), 0, np.ascontiguousarray(i.values),
                          i.valid_bitmap(), 0, i.length, True)
    got = to_list(vals, bm, i.length)
    assert got == case["want"]
    assert vc == sum(x is not None for x in case["want"])
    # null slots are zero-filled (WriteZero, gather_internal.h:114-153)
    assert all(vals[k] == 0 for k, x in enumerate(case["want"]) if x is None)


@pytest.mark.parametrize("case", GOLD["take_index_error"], ids=lambda c: c["cite"])
def test_golden_take_index_error(case):
    i = from_list(case["indices"], np.int8)
    bad = O.check_index_bounds(np.ascontiguousarray(i.values), None, 0, i.length, len(case["values"]))
    assert bad == case["bad"]


def test_golden_cast():
    c = GOLD["cast_f64_f32"]
    assert O.cast_f64_f32(np.array(c["values"])).tolist() == [float(np.float32(x)) for x in c["want"]]


# ------------------------------------------------------------------ 2. the reference's own build
needs_pa = pytest.mark.skipif(pa is None, reason="pyarrow wheel not importable")


def rng_for(*key):
    return np.random.default_rng([U.kRandomSeed, *[abs(hash(str(k))) % (1 << 31) for k in key]])


def pa_to_list(a):
    return a.to_pylist()


@needs_pa
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("null_p", [0.0, 0.01, 0.1, 0.999, 1.0])
@pytest.mark.parametrize("true_p", [0.0, 0.1, 0.999, 1.0])
def test_filter_vs_pyarrow_random_grid(true_p, null_p, sel):
    """FilterRandomTest (vector_selection_test.cc:2241-2260): len 1024(+), differing offsets."""
    rng = rng_for("pf", true_p, null_p, sel)
    for dtype, voff, moff in ((np.int64, 0, 0), (np.int32, 3, 5), (np.int8, 1, 64), (np.float64, 7, 2)):
        v = U.random_array(rng, dtype, 1500, null_p=null_p, offset=voff, tail=2)
        m = U.random_mask(rng, 1500, true_p, null_p=null_p / 2, offset=moff, tail=1)
        ref = pc.filter(v.to_pyarrow(), m.to_pyarrow(), null_selection_behavior=sel)
        got = o_filter(v, m, sel)
        rl = ref.to_pylist()
        if np.dtype(dtype).kind == "f":
            assert len(got) == len(rl)
            for g, r in zip(got, rl):
                assert (g is None) == (r is None) and (g is None or g == r or (g != g and r != r))
        else:
            assert got == rl


@needs_pa
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
def test_mask_to_indices_vs_pyarrow(sel):
    """indices_nonzero / filter(iota) give the reference's GetTakeIndices result."""
    rng = rng_for("pm2i", sel)
    for n, off in ((0, 0), (1, 0), (1000, 3), (70000, 1)):
        m = U.random_mask(rng, n, 0.3, null_p=0.1, offset=off)
        code = 1 if sel == "emit_null" else 0
        vals, bm = O.mask_to_indices(m.data_bytes(), m.valid_bitmap(), m.offset, n, code, True)
        iota = pa.array(np.arange(n, dtype=vals.dtype))
        ref = pc.filter(iota, m.to_pyarrow(), null_selection_behavior=sel)
        assert to_list(vals, bm, len(vals)) == ref.to_pylist()


@needs_pa
@pytest.mark.parametrize("idx_dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32,
                                       np.uint64, np.int64])
@pytest.mark.parametrize("null_p", [0.0, 0.01, 0.1, 0.5, 1.0])
def test_take_vs_pyarrow_random(idx_dtype, null_p):
    """TakeRandomTest (vector_selection_test.cc:2282-2315): values 1025 (127 for 8-bit), indices 257."""
    rng = rng_for("pt", idx_dtype, null_p)
    nv = 127 if np.dtype(idx_dtype).itemsize == 1 else 1025
    for dtype in (np.int64, np.int16):
        v = U.random_array(rng, dtype, nv, null_p=null_p, offset=2)
        i = U.random_array(rng, idx_dtype, 257, null_p=null_p / 2, offset=1, lo=0, hi=nv - 1)
        vals, bm, vc = O.take(v.data_bytes(), v.valid_bitmap(), v.offset, np.ascontiguousarray(i.values),
                              i.valid_bitmap(), i.offset, i.length, True)
        ref = pc.take(v.to_pyarrow(), i.to_pyarrow())
        assert to_list(vals, bm, i.length) == ref.to_pylist()
        assert ref.null_count == i.length - vc


@needs_pa
def test_bounds_message_vs_pyarrow():
    v = pa.array(np.arange(10))
    for idx in ([0, 10, 3, -4], [0, -1, 12], [2**40, 1]):
        i = np.array(idx, dtype=np.int64)
        bad = O.check_index_bounds(i, None, 0, len(i), 10)
        with pytest.raises(pa.lib.ArrowIndexError) as e:
            pc.take(v, pa.array(i))
        assert str(e.value) == f"Index {bad} out of bounds"


@needs_pa
def test_cast_and_compare_vs_pyarrow():
    rng = rng_for("pcast")
    x = rng.standard_normal(20000)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e39, 1e-46, 3.4028235677973366e38]
    x[100:2000] *= 1e40
    x[2000:4000] *= 1e-42
    got = O.cast_f64_f32(x)
    ref = pc.cast(pa.array(x), pa.float32(), safe=False).to_numpy()
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all()
    y = rng.standard_normal(20000)
    y[::5] = x[::5]
    bits = O.unpack_bits(O.greater_f64(x, y), 0, len(x))
    assert (bits == pc.greater(pa.array(x), pa.array(y)).to_numpy(zero_copy_only=False)).all()
    assert (O.unpack_bits(O.greater_f64(x, 0.5), 0, len(x))
            == pc.greater(pa.array(x), pa.scalar(0.5)).to_numpy(zero_copy_only=False)).all()
    a = rng.integers(-2**63, 2**63 - 1, size=5000, dtype=np.int64)
    b = rng.integers(-2**63, 2**63 - 1, size=5000, dtype=np.int64)
    assert (O.add(a, b) == pc.add(pa.array(a), pa.array(b)).to_numpy()).all()  # unchecked add wraps


@needs_pa
@pytest.mark.parametrize("order", ["ascending", "descending"])
@pytest.mark.parametrize("placement", ["at_end", "at_start"])
@pytest.mark.parametrize("dtype,lo,hi,n", [(np.uint64, 0, 5, 700), (np.uint64, None, None, 3000),
                                            (np.int64, -3, 3, 2000), (np.int64, None, None, 3000)])
def test_sort_indices_vs_pyarrow(dtype, lo, hi, n, order, placement):
    """Both the counting-sort (small range, len >= 1024) and the stable_sort branch
    (vector_array_sort.cc:404-446) must give the oracle's permutation."""
    rng = rng_for("psort", dtype, lo, n, order, placement)
    a = U.random_array(rng, dtype, n, null_p=0.15, offset=3, lo=lo, hi=hi)
    got = O.sort_indices_64(np.ascontiguousarray(a.values), a.valid_bitmap(), a.offset, n,
                            descending=(order == "descending"), nulls_at_start=(placement == "at_start"))
    ref = pc.array_sort_indices(a.to_pyarrow(), order=order, null_placement=placement).to_numpy()
    assert (got == ref).all()


@needs_pa
@pytest.mark.parametrize("skip_nulls,min_count", [(True, 1), (False, 1), (True, 0), (True, 4), (False, 0)])
@pytest.mark.parametrize("use_threads", [False, True])
def test_groupby_sum_vs_pyarrow(skip_nulls, min_count, use_threads):
    """RunGroupBy with use_threads in {false,true}, key-sorted comparison
    (acero/hash_aggregate_test.cc:262-280, 398-412)."""
    rng = rng_for("pgb", skip_nulls, min_count)
    n = 20000
    k = U.random_array(rng, np.int32, n, null_p=0.05, offset=2, lo=-50, hi=50)
    v = U.random_array(rng, np.int64, n, null_p=0.2, offset=1)
    w = O.groupby_sum_i64(np.ascontiguousarray(k.values), k.valid_bitmap(), k.offset,
                          np.ascontiguousarray(v.values), v.valid_bitmap(), v.offset, n,
                          skip_nulls, min_count)
    got = sorted(((0 if kv else 1, int(kk) if kv else 0, int(s) if ok else None)
                  for kk, kv, s, ok in zip(w["keys"], w["key_is_valid"], w["sums"], w["valid"])),
                 key=lambda r: (r[0], r[1]))
    t = pa.table({"k": k.to_pyarrow(), "v": v.to_pyarrow()})
    r = t.group_by("k", use_threads=use_threads).aggregate(
        [("v", "sum", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=min_count))])
    ref = sorted(((1 if kk is None else 0, 0 if kk is None else kk, s)
                  for kk, s in zip(r.column("k").to_pylist(), r.column("v_sum").to_pylist())),
                 key=lambda r_: (r_[0], r_[1]))
    assert got == ref


# ------------------------------------------------------------------ 3. committed pyarrow fixtures
def test_oracle_vs_committed_pyarrow_fixtures():
    path = os.path.join(HERE, "golden", "pyarrow_golden.npz")
    z = np.load(path)
    n = int(z["n"])
    vals, valid, mask, mvalid = z["values"], z["values_valid"], z["mask"], z["mask_valid"]
    v = U.HostArray(vals, valid, 0, n)
    m = U.HostArray(mask, mvalid, 0, n)
    for sel in ("drop", "emit_null"):
        code = 1 if sel == "emit_null" else 0
        ov, obm = O.filter(v.data_bytes(), v.valid_bitmap(), 0, m.data_bytes(), m.valid_bitmap(), 0, n,
                           code, True)
        ovalid = O.unpack_bits(obm, 0, len(ov))
        assert (ovalid == z[f"filter_{sel}_valid"]).all()
        assert (ov[ovalid] == z[f"filter_{sel}_values"][ovalid]).all()
    idx, ivalid = z["indices"], z["indices_valid"]
    tv, tbm, _ = O.take(v.data_bytes(), v.valid_bitmap(), 0, idx, O.pack_bits(ivalid), 0, len(idx), True)
    tvalid = O.unpack_bits(tbm, 0, len(idx))
    assert (tvalid == z["take_valid"]).all() and (tv[tvalid] == z["take_values"][tvalid]).all()
    assert (O.cast_f64_f32(z["f64"]).view(np.uint32) == z["cast_f32"].view(np.uint32)).all()
    assert (O.unpack_bits(O.greater_f64(z["f64"], z["f64_b"]), 0, len(z["f64"])) == z["greater"]).all()
    for order in ("ascending", "descending"):
        for placement in ("at_end", "at_start"):
            got = O.sort_indices_64(z["sort_keys"], O.pack_bits(z["sort_valid"]), 0, len(z["sort_keys"]),
                                    descending=(order == "descending"),
                                    nulls_at_start=(placement == "at_start"))
            assert (got == z[f"sort_{order}_{placement}"]).all()
    w = O.groupby_sum_i64(z["gb_keys"], O.pack_bits(z["gb_keys_valid"]), 0, z["gb_vals"],
                          O.pack_bits(z["gb_vals_valid"]), 0, len(z["gb_keys"]))
    got = sorted(((0 if kv else 1, int(kk) if kv else 0, int(s) if ok else None)
                  for kk, kv, s, ok in zip(w["keys"], w["key_is_valid"], w["sums"], w["valid"])),
                 key=lambda r: (r[0], r[1]))
    want = sorted(((int(a), int(b), None if not c else int(d))
                   for a, b, c, d in zip(z["gb_ref_isnull"], z["gb_ref_key"], z["gb_ref_valid"],
                                         z["gb_ref_sum"])), key=lambda r: (r[0], r[1]))
    assert got == want


# ------------------------------------------------------------------ sort: other key types
@pytest.mark.parametrize("name", ["u32", "i32", "f64", "f32"])
def test_golden_typed_sort_fixture(name):
    """Fixtures generated from the reference build (tests/golden/make_golden.py): 32-bit and
    floating point keys, NaNs next to the nulls whatever the order, -0.0 tying with 0.0."""
    z = np.load(os.path.join(HERE, "golden", "pyarrow_golden.npz"))
    keys, valid = z[f"tsort_{name}_keys"], z[f"tsort_{name}_valid"]
    for order in ("ascending", "descending"):
        for placement in ("at_end", "at_start"):
            got = O.sort_indices(np.ascontiguousarray(keys), O.pack_bits(valid), 0, len(keys),
                                 descending=(order == "descending"), nulls_at_start=(placement == "at_start"))
            assert (got == z[f"tsort_{name}_{order}_{placement}"]).all(), (name, order, placement)


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not importable")
@pytest.mark.parametrize("dtype", [np.uint32, np.int32, np.float64, np.float32, np.uint64, np.int64])
def test_typed_sort_vs_pyarrow(dtype):
    rng = rng_for("tsortpin", str(dtype))
    n = 20000
    a = U.random_array(rng, dtype, n, null_p=0.07, offset=5)
    if np.dtype(dtype).kind == "f":
        a.values[::5] = np.nan
        a.values[::9] = -0.0
        a.values[1::9] = 0.0
        a.values[::4] = np.round(a.values[::4])
    else:
        a.values[::4] = a.values[::4] % 21
    for order in ("ascending", "descending"):
        for placement in ("at_end", "at_start"):
            got = O.sort_indices(np.ascontiguousarray(a.values), a.valid_bitmap(), a.offset, n,
                                 descending=(order == "descending"), nulls_at_start=(placement == "at_start"))
            ref = pc.array_sort_indices(a.to_pyarrow(), order=order, null_placement=placement).to_numpy()
            assert (got == ref).all(), (dtype, order, placement)


# ------------------------------------------------------------------ binary / utf8 take + filter
def _bin_from_list(xs, utf8=True):
    valid = np.array([x is not None for x in xs], dtype=bool)
    enc = [b"" if x is None else (x.encode() if isinstance(x, str) else x) for x in xs]
    offsets = np.zeros(len(xs) + 1, dtype=np.int32)
    np.cumsum([len(e) for e in enc], out=offsets[1:])
    data = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
    return U.HostBinaryArray(offsets, data, None if valid.all() else valid, 0, len(xs), utf8)


def _bin_to_list(res, n):
    off, data, bm, _vc = res
    valid = O.unpack_bits(bm, 0, n)
    raw = data.tobytes()
    return [raw[off[i]: off[i + 1]].decode() if valid[i] else None for i in range(n)]


# TYPED_TEST(TestTakeKernelWithString, TakeString) vector_selection_test.cc:1661-1664
@pytest.mark.parametrize("values,indices,want", [
    (["a", "b", "c"], [0, 1, 0], ["a", "b", "a"]),
    ([None, "b", "c"], [0, 1, 0], [None, "b", None]),
    (["a", "b", "c"], [None, 1, 0], [None, "b", "a"]),
])
def test_golden_binary_take(values, indices, want):
    v, i = _bin_from_list(values), from_list(indices, np.int32)
    res = O.binary_take(v.offsets, v.data, v.valid_bitmap(), 0, i.values, i.valid_bitmap(), 0, i.length)
    assert _bin_to_list(res, i.length) == want


# TYPED_TEST(TestFilterKernelWithString, FilterString) vector_selection_test.cc:670-674 (EMIT_NULL is
# the default of the test fixture's AssertFilter; DROP is checked alongside, :117-160)
@pytest.mark.parametrize("values,mask,want_emit,want_drop", [
    (["a", "b", "c"], [0, 1, 0], ["b"], ["b"]),
    ([None, "b", "c"], [0, 1, 0], ["b"], ["b"]),
    (["a", "b", "c"], [None, 1, 0], [None, "b"], ["b"]),
])
def test_golden_binary_filter(values, mask, want_emit, want_drop):
    v, m = _bin_from_list(values), from_list(mask, np.bool_)
    for code, want in ((1, want_emit), (0, want_drop)):
        res = O.binary_filter(v.offsets, v.data, v.valid_bitmap(), 0, m.data_bytes(), m.valid_bitmap(), 0,
                              m.length, code)
        assert _bin_to_list(res, len(res[0]) - 1) == want


def _assert_bin_equals_pyarrow(res, ref):
    off, data, bm, vc = res
    n = len(off) - 1
    got = pa.Array.from_buffers(ref.type, n, [pa.py_buffer(bm.tobytes()), pa.py_buffer(off.tobytes()),
                                              pa.py_buffer(data.tobytes())])
    assert got.equals(ref)
    assert n - vc == ref.null_count
    roff = np.frombuffer(ref.buffers()[1], dtype=np.int32)[ref.offset: ref.offset + n + 1]
    assert (off == roff - roff[0]).all()


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("utf8", [True, False])
@pytest.mark.parametrize("vnull,inull", [(0.0, 0.0), (0.2, 0.0), (0.0, 0.1), (0.5, 0.5)])
def test_binary_take_vs_pyarrow(utf8, vnull, inull):
    rng = np.random.default_rng([U.kRandomSeed, int(utf8), int(vnull * 100), int(inull * 100)])
    v = U.random_binary(rng, 3000, null_p=vnull, offset=7, tail=3, utf8=utf8)
    i = U.random_array(rng, np.int32, 10_000, null_p=inull, offset=2, lo=0, hi=2999)
    res = O.binary_take(v.offsets, v.data, v.valid_bitmap(), v.offset, i.values, i.valid_bitmap(), i.offset,
                        i.length)
    _assert_bin_equals_pyarrow(res, pc.take(v.to_pyarrow(), i.to_pyarrow()))


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("true_p,vnull,mnull", [(0.0, 0.1, 0.0), (0.3, 0.0, 0.0), (0.5, 0.2, 0.1), (1.0, 0.1, 0.05)])
def test_binary_filter_vs_pyarrow(sel, true_p, vnull, mnull):
    rng = np.random.default_rng([U.kRandomSeed, int(true_p * 100), int(vnull * 100), int(mnull * 100)])
    n = 20_000
    v = U.random_binary(rng, n, null_p=vnull, offset=5, tail=3)
    m = U.random_mask(rng, n, true_p, null_p=mnull, offset=2, tail=1)
    res = O.binary_filter(v.offsets, v.data, v.valid_bitmap(), v.offset, m.data_bytes(), m.valid_bitmap(),
                          m.offset, m.length, 1 if sel == "emit_null" else 0)
    _assert_bin_equals_pyarrow(res, pc.filter(v.to_pyarrow(), m.to_pyarrow(), null_selection_behavior=sel))


# ------------------------------------------------------------------ hash_min_max (oracle.groupby_minmax_i64)
def test_golden_groupby_min_max():
    """TEST_P(GroupBy, MinMaxOnly), acero/hash_aggregate_test.cc:1591-1660, first aggregate: the
    float64 arguments scaled by 8 (all exactly representable, order preserved) so that they are int64;
    keys 1, 2, 3 and the null key; key 3 has only null values -> null min/max."""
    arg = [8, None, 0, None, 32, 26, 1, -2, 6, None]
    key = [1, 1, 2, 3, None, 1, 2, 2, None, 3]
    v, k = from_list(arg, np.int64), from_list(key, np.int32)
    w = O.groupby_minmax_i64(k.values, k.valid_bitmap(), 0, v.values, v.valid_bitmap(), 0, len(arg), True)
    got = {(int(a) if b else None): ((int(c), int(d)) if e else None)
           for a, b, c, d, e in zip(w["keys"], w["key_is_valid"], w["mins"], w["maxs"], w["valid"])}
    assert got == {1: (8, 26), 2: (-2, 1), 3: None, None: (6, 32)}


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("skip_nulls", [True, False])
@pytest.mark.parametrize("knull,vnull", [(0.0, 0.0), (0.05, 0.3), (0.2, 0.95)])
def test_groupby_min_max_vs_pyarrow(skip_nulls, knull, vnull):
    rng = np.random.default_rng([U.kRandomSeed, int(skip_nulls), int(knull * 100), int(vnull * 100)])
    n = 20_000
    k = U.random_array(rng, np.int32, n, null_p=knull, offset=3, lo=-300, hi=300)
    v = U.random_array(rng, np.int64, n, null_p=vnull, offset=1)
    w = O.groupby_minmax_i64(k.values, k.valid_bitmap(), k.offset, v.values, v.valid_bitmap(), v.offset, n, skip_nulls)
    got = {(int(a) if b else None): ((int(c), int(d)) if e else None)
           for a, b, c, d, e in zip(w["keys"], w["key_is_valid"], w["mins"], w["maxs"], w["valid"])}
    t = pa.table({"k": k.to_pyarrow(), "v": v.to_pyarrow()})
    r = t.group_by("k", use_threads=False).aggregate(
        [("v", "min_max", pc.ScalarAggregateOptions(skip_nulls=skip_nulls, min_count=5))])   # min_count is ignored
    ref = {a: (None if b is None or b["min"] is None else (b["min"], b["max"]))
           for a, b in zip(r.column("k").to_pylist(), r.column("v_min_max").to_pylist())}
    assert got == ref


# ------------------------------------------------------------------ unique / value_counts (oracle.unique_i32)
# TYPED_TEST(TestHashKernelPrimitive, Unique / ValueCounts), kernels/vector_hash_test.cc:159-186,188-215
@pytest.mark.parametrize("values,want", [
    ([2, None, 2, 1], [2, None, 1]),
    ([None, None, 3, 1], [None, 3, 1]),
    ([2, None, 3, 2], [2, None, 3]),            # [1, 2, null, 3, 2, null].Slice(1, 4)
])
def test_golden_unique(values, want):
    a = from_list(values, np.int32)
    v, ok = O.unique_i32(a.values, a.valid_bitmap(), 0, a.length)
    assert [int(x) if y else None for x, y in zip(v, ok)] == want


def test_golden_value_counts():
    a = from_list([2, None, 2, 1, 2, 3, None], np.int32)     # vector_hash_test.cc:206-211
    v, ok, c = O.unique_i32(a.values, a.valid_bitmap(), 0, a.length, True)
    assert [int(x) if y else None for x, y in zip(v, ok)] == [2, None, 1, 3] and c.tolist() == [3, 2, 1, 1]


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("null_p", [0.0, 0.1, 1.0])
def test_unique_vs_pyarrow(null_p):
    rng = np.random.default_rng([U.kRandomSeed, int(null_p * 100)])
    a = U.random_array(rng, np.int32, 30_000, null_p=null_p, offset=5, lo=-500, hi=500)
    v, ok, c = O.unique_i32(a.values, a.valid_bitmap(), a.offset, a.length, True)
    ref = pc.value_counts(a.to_pyarrow())
    assert [int(x) if y else None for x, y in zip(v, ok)] == ref.field("values").to_pylist()
    assert c.tolist() == ref.field("counts").to_pylist()


# ------------------------------------------------------------------ and_kleene / or_kleene (oracle.kleene)
def test_golden_kleene_truth_table():
    """TEST(TestBooleanKernel, KleeneAnd / KleeneOr), kernels/scalar_boolean_test.cc: the 3x3 table."""
    tf = [True, True, True, False, False, False, None, None, None]
    ft = [True, False, None, True, False, None, True, False, None]
    l, r = from_list(tf, np.bool_), from_list(ft, np.bool_)
    lv = None if l.valid is None else l.logical_valid()
    rv = None if r.valid is None else r.logical_valid()
    for op, want in (("and", [True, False, None, False, False, False, None, False, None]),
                     ("or", [True, True, True, True, False, None, True, None, None])):
        data, valid = O.kleene(op, l.logical_values(), lv, r.logical_values(), rv)
        assert [bool(d) if v else None for d, v in zip(data, valid)] == want, op


@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("lnull,rnull", [(0.0, 0.0), (0.3, 0.0), (0.3, 0.4)])
def test_kleene_vs_pyarrow(lnull, rnull):
    rng = np.random.default_rng([U.kRandomSeed, int(lnull * 10), int(rnull * 10)])
    l = U.random_mask(rng, 10_000, 0.5, null_p=lnull, offset=3)
    r = U.random_mask(rng, 10_000, 0.5, null_p=rnull, offset=70)
    lv = None if l.valid is None else l.logical_valid()
    rv = None if r.valid is None else r.logical_valid()
    for op, fn in (("and", pc.and_kleene), ("or", pc.or_kleene)):
        data, valid = O.kleene(op, l.logical_values(), lv, r.logical_values(), rv)
        ref = fn(l.to_pyarrow(), r.to_pyarrow())
        assert [bool(d) if v else None for d, v in zip(data, valid)] == ref.to_pylist()


# ------------------------------------------------------------------ the comparison family (oracle.compare)
@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("op", ["equal", "not_equal", "greater", "greater_equal", "less", "less_equal"])
def test_compare_family_vs_pyarrow(op):
    """Values follow TestCompareKernel's SimpleCompare cases (kernels/scalar_compare_test.cc): ties,
    NaN on either side, signed zeros, infinities; int64 edges."""
    f = np.array([0.0, -0.0, 1.5, np.nan, np.inf, -np.inf, 2.0, np.nan, -1.0, 1.5])
    g = np.array([-0.0, 0.0, 1.5, 1.0, np.inf, np.inf, np.nan, np.nan, -2.0, 1.25])
    i = np.array([0, -1, 2**63 - 1, -2**63, 5, 5, 7], dtype=np.int64)
    j = np.array([0, 1, -2**63, 2**63 - 1, 5, 6, 7], dtype=np.int64)
    for l, r in ((f, g), (i, j), (f, 1.5), (1.5, g), (i, 5), (5, j)):
        pl = pa.array(l) if isinstance(l, np.ndarray) else pa.scalar(l)
        pr = pa.array(r) if isinstance(r, np.ndarray) else pa.scalar(r)
        assert O.compare(op, l, r).tolist() == getattr(pc, op)(pl, pr).to_pylist(), (op, l, r)


# ------------------------------------------------------------------ add / subtract / multiply (+ checked) (oracle.arith)
@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("op", ["add", "subtract", "multiply"])
def test_arith_vs_pyarrow(op):
    i = np.array([2**63 - 1, -2**63, 2**62, 5, -7, 0], dtype=np.int64)
    j = np.array([1, -1, 2, 3, 4, 9], dtype=np.int64)
    if op == "subtract":
        j = -j
    want, ovf = O.arith(op, i, j)
    assert ovf and getattr(pc, op)(pa.array(i), pa.array(j)).to_pylist() == want.tolist()      # unchecked wraps
    with pytest.raises(pa.lib.ArrowInvalid, match="overflow"):
        getattr(pc, op + "_checked")(pa.array(i), pa.array(j))
    valid = np.array([False, False, False, True, True, True])       # the overflowing slots are null: no error
    want, ovf = O.arith(op, i, j, valid)
    ref = getattr(pc, op + "_checked")(pa.array(i, mask=~valid), pa.array(j))
    assert not ovf and ref.to_pylist()[3:] == want[3:].tolist()
    f, g = np.array([1.5, np.inf, -0.0, 1e308]), np.array([2.25, -np.inf, 0.0, 1e308])
    want, _ = O.arith(op, f, g)
    got = getattr(pc, op + "_checked")(pa.array(f), pa.array(g)).to_numpy()
    assert np.array_equal(want, got, equal_nan=True)


# ------------------------------------------------------------------ dictionary_encode (oracle.dictionary_encode_i32)
@pytest.mark.skipif(pc is None, reason="pyarrow wheel not in this image")
@pytest.mark.parametrize("mode", ["mask", "encode"])
def test_dictionary_encode_vs_pyarrow(mode):
    """TYPED_TEST(TestHashKernelPrimitive, DictEncode) shape (kernels/vector_hash_test.cc): {2, 1, 2, 1, 2, 3}
    with a null in the middle, then a random grid."""
    rng = np.random.default_rng([U.kRandomSeed, len(mode)])
    for a in (from_list([2, 1, None, 1, 2, 3, None], np.int32), U.random_array(rng, np.int32, 20_000, null_p=0.1, offset=4, lo=-90, hi=90)):
        idx, iv, dv, dvv = O.dictionary_encode_i32(a.values, a.valid_bitmap(), a.offset, a.length, mode == "encode")
        ref = pc.dictionary_encode(a.to_pyarrow(), null_encoding=mode)
        assert [int(x) if ok else None for x, ok in zip(idx, iv)] == ref.indices.to_pylist()
        assert [int(x) if ok else None for x, ok in zip(dv, dvv)] == ref.dictionary.to_pylist()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sort_indices_multi_vs_pyarrow(seed):
    """The numpy restatement of the multi-key table sorter against the reference build: random direction and
    null placement per key, few distinct values (ties everywhere), NaNs, chunked columns."""
    rng = np.random.default_rng(seed)
    n = 5000
    keys, cols = [], {}
    for j, dt in enumerate((np.int32, np.float64, np.int64)):
        v = rng.integers(-3, 3, n).astype(dt)
        if dt is np.float64:
            v = np.where(rng.random(n) < 0.1, np.nan, v / 2)
            v[rng.random(n) < 0.05] = -0.0
        valid = rng.random(n) >= 0.15
        keys.append((v, valid))
        arr = pa.array(v, mask=~valid)
        cols[f"k{j}"] = pa.chunked_array([arr.slice(0, 1234), arr.slice(1234)])
    desc = [bool(b) for b in rng.integers(0, 2, 3)]
    start = [bool(b) for b in rng.integers(0, 2, 3)]
    want = pc.sort_indices(pa.table(cols), sort_keys=[(f"k{j}", "descending" if desc[j] else "ascending",
                                                       "at_start" if start[j] else "at_end") for j in range(3)])
    got = O.sort_indices_multi(keys, desc, start)
    assert np.array_equal(got, want.to_numpy())
    assert np.array_equal(O.sort_indices_multi(keys[:1], desc[:1], start[0]),
                          pc.sort_indices(pa.table(cols), sort_keys=[("k0", "descending" if desc[0] else "ascending",
                                                                      "at_start" if start[0] else "at_end")]).to_numpy())


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.int32, np.uint32, np.float64, np.float32])
@pytest.mark.parametrize("case", GOLD["sort_indices_integral"] + GOLD["sort_indices_real"],
                         ids=lambda c: f"{c['cite'].split(' ')[0]}-{c['order']}-{c['null_placement']}-{c['values']}"[:90])
def test_golden_sort_indices(case, dtype):
    """vector_sort_test.cc:640-724: stability on ties, descending ties, null placement, NaNs between the values and
    the nulls — the oracle's argsort equals the expected permutation for every key type the path registers."""
    a = golden_sort_array(case["values"], dtype)
    if a is None:
        pytest.skip("values of this case do not fit the key type")
    got = O.sort_indices(np.ascontiguousarray(a.values), a.valid_bitmap(), a.offset, a.length,
                         descending=case["order"] == "descending", nulls_at_start=case["null_placement"] == "at_start")
    assert got.tolist() == case["want"]


@pytest.mark.parametrize("name", ["uint8", "int8", "int64"])
def test_golden_sort_indices_of_the_8_bit_extremes_and_a_wide_int64_range(name):
    """vector_sort_test.cc:839-885 (SortUInt8 / SortInt8 / SortInt64) on the oracle's argsort; the 8-bit keys as the
    device route sorts them — widened to 32 bits."""
    for case in GOLD["sort_indices_narrow_and_wide"][name]:
        a = golden_sort_array(case["values"], {"uint8": np.uint32, "int8": np.int32, "int64": np.int64}[name])
        got = O.sort_indices(np.ascontiguousarray(a.values), a.valid_bitmap(), a.offset, a.length,
                             descending=case["order"] == "descending", nulls_at_start=case["null_placement"] == "at_start")
        assert got.tolist() == case["want"], case


def test_golden_hash_sum_sum_only():
    """acero/hash_aggregate_test.cc:839-883 (SumOnly): three batches, null key = its own group, all-null group -> null."""
    g = GOLD["hash_sum_sum_only"]
    keys = [k for b in g["batches"] for k in b["key"]]
    vals = [v for b in g["batches"] for v in b["argument"]]
    k, v = from_list(keys, np.int32), from_list(vals, np.int64)
    w = O.groupby_sum_i64(np.ascontiguousarray(k.values), k.valid_bitmap(), 0, np.ascontiguousarray(v.values), v.valid_bitmap(),
                          0, len(keys))
    rows = sorted(((int(a) if b else None, int(c) if d else None) for a, b, c, d in
                   zip(w["keys"], w["key_is_valid"], w["sums"], w["valid"])), key=lambda r: (r[0] is None, r[0] or 0))
    assert [list(r) for r in rows] == g["want_sorted_by_key"]


@pytest.mark.parametrize("section", ["grouper_numeric_key", "grouper_floating_point_key", "grouper_multiple_int_keys"])
def test_golden_grouper(section):
    """compute/row/grouper_test.cc:845-910, :1024-1057 — the reference's own Grouper expectations (exact ids, uniques,
    Lookup nulls) replayed on the oracle's restatement of GrouperImpl."""
    from tests import parity_cases as P

    assert P.replay_golden_grouper(GOLD, section, P.oracle_grouper_factory) > 0


def test_grouper_vectorised_oracle_equals_the_row_at_a_time_one():
    """grouper_ids_one_batch (np.unique based, what the large parity cases use) against the dict restatement, and the
    per-key sums of both against pyarrow's Table.group_by on two key columns with nulls."""
    rng = np.random.default_rng(77)
    for dtypes, n, card, null_p in (((np.int64,), 3000, 40, 0.1), ((np.int32, np.int16), 5000, 300, 0.2),
                                    ((np.uint8, np.int64, np.uint16), 2000, 1500, 0.05), ((np.float64,), 1000, 12, 0.3)):
        cols = []
        for dt in dtypes:
            pool = (rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, size=card, dtype=dt, endpoint=True)
                    if np.issubdtype(dt, np.integer) else rng.standard_normal(card).astype(dt))
            cols.append((pool[rng.integers(0, card, size=n)], rng.random(n) >= null_p))
        ids, first = O.grouper_ids_one_batch(cols)
        slow = O.Grouper(len(dtypes))
        want = slow.consume(cols)
        assert np.array_equal(ids, want)
        assert np.array_equal(first, [int(np.flatnonzero(want == g)[0]) for g in range(slow.num_groups)])
    if pa is not None:
        cols = [(rng.integers(-3, 3, size=4000, dtype=np.int64), rng.random(4000) >= 0.1),
                (rng.integers(0, 4, size=4000, dtype=np.int32), rng.random(4000) >= 0.1)]
        v = rng.integers(-1000, 1000, size=4000, dtype=np.int64)
        ids, first = O.grouper_ids_one_batch(cols)
        sums = np.zeros(len(first), dtype=np.int64)
        np.add.at(sums, ids, v)
        t = pa.table({"a": pa.array(cols[0][0], mask=~cols[0][1]), "b": pa.array(cols[1][0], mask=~cols[1][1]), "v": v})
        ref = t.group_by(["a", "b"], use_threads=False).aggregate([("v", "sum")]).to_pydict()
        ref_map = {(a, b): s for a, b, s in zip(ref["a"], ref["b"], ref["v_sum"])}
        got_map = {(int(cols[0][0][r]) if cols[0][1][r] else None, int(cols[1][0][r]) if cols[1][1][r] else None): int(sums[g])
                   for g, r in enumerate(first)}
        assert got_map == ref_map


def _reference_grouper(cols, batch_rows):
    """The reference's own Grouper (GrouperFastImpl in the wheel's libarrow_compute) on (values, valid) columns through
    oracle/ref/grouper_probe.cc; returns (ids, [(unique values, valid)])."""
    import subprocess
    import tempfile

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.dirname(pa.__file__)
    src = os.path.join(ROOT, "oracle", "ref", "grouper_probe.cc")
    exe = os.path.join(ROOT, "oracle", "_build", "grouper_probe")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        libs = [os.path.join(d, f) for f in sorted(os.listdir(d))
                if f.startswith(("libarrow.so.", "libarrow_compute.so.")) and f.count(".") == 2]
        tmp = f"{exe}.tmp{os.getpid()}"
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-I", os.path.join(d, "include"), src, "-o", tmp, *libs,
                               f"-Wl,-rpath,{d}"])
        os.replace(tmp, exe)
    n = len(cols[0][0])
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([n, len(cols), batch_rows], dtype=np.int64).tobytes())
            for values, valid in cols:
                f.write(np.array([values.dtype.itemsize, int(values.dtype.kind == "f")], dtype=np.int64).tobytes())
                f.write(np.ascontiguousarray(values).tobytes())
                f.write(np.ascontiguousarray(valid, dtype=np.uint8).tobytes())
        subprocess.check_call([exe, fin, fout])
        raw = open(fout, "rb").read()
    g = int(np.frombuffer(raw, dtype=np.int64, count=1)[0])
    ids = np.frombuffer(raw, dtype=np.uint32, count=n, offset=8)
    off = 8 + 4 * n
    uniq = []
    for values, _ in cols:
        w = values.dtype.itemsize
        uv = np.frombuffer(raw, dtype=values.dtype, count=g, offset=off)
        off += g * w
        uvalid = np.frombuffer(raw, dtype=np.uint8, count=g, offset=off).astype(bool)
        off += g
        uniq.append((uv, uvalid))
    return ids, uniq


@pytest.mark.skipif(pa is None, reason="needs the pyarrow wheel (the reference build)")
def test_oracle_grouper_against_the_reference_grouper_itself():
    """oracle.Grouper vs arrow::compute::Grouper from the wheel on random key rows (1-4 fixed-width columns, nulls, several
    Consume calls): the same partition of the rows into groups (ids equal up to a bijection — all that the reference's
    own AssertEquivalentIds asks, grouper_test.cc:676-714), the same set of unique rows, uniques[id] = the row's key on
    both sides, and — what the device Grouper additionally promises — whether the reference's ids are in order of first
    appearance (they are for the generic GrouperImpl; the fast implementation hands out ids per 1024-row minibatch in
    hash order, so only small batches show it)."""
    rng = np.random.default_rng(2024)
    cases = [((np.int64,), 5000, 300, 0.1, 5000), ((np.int32, np.int32), 20000, 700, 0.1, 4096),
             ((np.uint8, np.int64, np.uint16), 8000, 5000, 0.05, 1000), ((np.float64, np.int32), 3000, 40, 0.2, 3000),
             ((np.int16, np.int16, np.int16, np.int16), 12000, 9, 0.3, 777), ((np.int64,), 900, 900, 0.0, 100)]
    first_appearance_seen = False
    for dtypes, n, card, null_p, batch_rows in cases:
        cols = []
        for dt in dtypes:
            pool = (rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, size=card, dtype=dt, endpoint=True)
                    if np.issubdtype(dt, np.integer) else rng.standard_normal(card).astype(dt))
            valid = rng.random(n) >= null_p
            vals = pool[rng.integers(0, card, size=n)]
            vals[~valid] = 0
            cols.append((vals, valid))
        # the probe declares unsigned / float types of the same width: the Grouper compares bytes either way
        as_bytes = [(v.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[v.dtype.itemsize])
                     if v.dtype.kind != "f" else v, valid) for v, valid in cols]
        ref_ids, ref_uniq = _reference_grouper(as_bytes, batch_rows)
        mine = O.Grouper(len(cols))
        my_ids = np.concatenate([mine.consume([(v[b:b + batch_rows], valid[b:b + batch_rows]) for v, valid in cols])
                                 for b in range(0, n, batch_rows)])
        g = int(ref_ids.max()) + 1
        assert mine.num_groups == g == len(ref_uniq[0][0]), (dtypes, mine.num_groups, g)
        # same partition: the pairs (my id, reference id) form a bijection
        pairs = np.unique(np.stack([my_ids.astype(np.int64), ref_ids.astype(np.int64)], axis=1), axis=0)
        assert len(pairs) == g and len(np.unique(pairs[:, 0])) == g and len(np.unique(pairs[:, 1])) == g, dtypes
        # uniques[id] is the row's key, on both sides
        my_uniq = mine.uniques([v.dtype for v, _ in cols])
        for (vals, valid), (rv, rvalid), (mv, mvalid) in zip(cols, ref_uniq, my_uniq):
            assert np.array_equal(rvalid[ref_ids], valid) and np.array_equal(mvalid[my_ids], valid)
            assert np.array_equal(rv.view(np.uint8).reshape(g, -1)[ref_ids][valid], vals.view(np.uint8).reshape(n, -1)[valid])
            assert np.array_equal(mv.view(np.uint8).reshape(g, -1)[my_ids][valid], vals.view(np.uint8).reshape(n, -1)[valid])
        first_appearance_seen |= bool(np.array_equal(my_ids, ref_ids))
    # the id ORDER: first appearance is what GrouperImpl produces and what every expectation of grouper_test.cc spells
    # out; the fast implementation (the one Make picks for these key types) agrees on inputs of that size ...
    tiny = [(np.array([3, 27, 3, 27, 0, 81, 27, 81], dtype=np.int64),
             np.array([1, 1, 1, 1, 0, 1, 1, 1], dtype=bool))]
    ref_ids, _ = _reference_grouper([(tiny[0][0].view(np.uint64), tiny[0][1])], 8)
    assert O.Grouper(1).consume(tiny).tolist() == ref_ids.tolist() == [0, 1, 0, 1, 2, 3, 1, 3]
    # ... and hands out ids in hash order inside its minibatches on larger ones (a bijection away, checked above)
    assert not first_appearance_seen


# ------------------------------------------------------------------ grouped aggregates: the reference's known answers
def _oracle_group_by(two_states=False):
    """golden_hash_aggregate.replay's `run` on the oracle: O.Grouper hands out dense ids batch by batch (one Grouper
    for the table), one oracle kernel state per aggregate consumes them.  two_states: batches alternate between two
    states per aggregate (each with its own Grouper) that are merged at the end through the Grouper mapping — the
    reference's "parallel/merged" leg (GroupByNode::Merge, acero/groupby_aggregate_node.cc)."""
    def run(table, aggs):
        batches = table.to_batches()
        key_dtype = table.schema.field("key").type.to_pandas_dtype()

        def make_states():
            out = []
            for col, fn, opts in aggs:
                skip = True if opts is None or fn == "count" else opts["skip_nulls"]
                minc = 1 if opts is None or fn == "count" else opts["min_count"]
                if fn == "count":
                    out.append(O.HashCountState("only_valid" if opts is None else opts["mode"]))
                elif fn == "count_all":
                    out.append(O.HashCountState("all"))
                elif fn == "sum":
                    out.append(O.HashSumState(skip, minc))
                elif fn in ("min", "max"):
                    out.append(O.HashMinMaxState(skip))
                elif fn in ("any", "all"):
                    out.append(O.HashBoolState(fn == "all", skip, minc))
                elif fn == "mean":
                    out.append({"rows": [], "skip": skip, "minc": minc})      # groupby_mean_i64 is whole-table: rows are kept
                else:
                    raise AssertionError(fn)
            return out

        lanes = [(O.Grouper(1), make_states())] + ([(O.Grouper(1), make_states())] if two_states else [])
        for bi, b in enumerate(batches):
            grouper, states = lanes[bi % len(lanes)]
            k = from_list(b.column("key").to_pylist(), key_dtype)
            ids = grouper.consume([(np.ascontiguousarray(k.values), k.valid if k.valid is not None else None)])
            for (col, fn, opts), st in zip(aggs, states):
                if fn == "count_all":
                    st.resize(grouper.num_groups)
                    st.consume(None, 0, ids)
                    continue
                xs = b.column(col).to_pylist()
                if fn in ("any", "all"):
                    a = from_list(xs, np.bool_)
                    st.resize(grouper.num_groups)
                    st.consume(np.asarray(a.values, bool), a.valid_bitmap(), 0, ids)
                elif fn == "count":
                    a = from_list([None if x is None else 0 for x in xs], np.int64)
                    st.resize(grouper.num_groups)
                    st.consume(a.valid_bitmap(), 0, ids)
                elif fn == "mean":
                    st["rows"] += list(zip(ids.tolist(), xs))
                else:
                    a = from_list(xs, np.int64)
                    st.resize(grouper.num_groups)
                    st.consume(np.ascontiguousarray(a.values), a.valid_bitmap(), 0, ids)
        grouper, states = lanes[0]
        if two_states:
            other_grouper, other_states = lanes[1]
            uniq = other_grouper.uniques([key_dtype])
            mapping = grouper.consume(uniq) if other_grouper.num_groups else np.zeros(0, np.uint32)
            for (col, fn, opts), st, ot in zip(aggs, states, other_states):
                if fn == "mean":
                    st["rows"] += [(int(mapping[g]), x) for g, x in ot["rows"]]     # (double sums are order-dependent only in the last ulp; these fixtures are exact)
                else:
                    st.resize(grouper.num_groups)
                    if ot.num_groups:
                        st.merge(ot, mapping)
        for st in states:
            if not isinstance(st, dict):
                st.resize(grouper.num_groups)
        (kv, kvalid), = grouper.uniques([key_dtype])
        keys = [int(v) if ok else None for v, ok in zip(kv, kvalid)]
        outs = []
        for (col, fn, opts), st in zip(aggs, states):
            if fn in ("count", "count_all"):
                outs.append(st.counts.tolist())
            elif fn == "sum":
                sums, valid, _ = st.finalize()
                outs.append([int(s) if ok else None for s, ok in zip(sums, valid)])
            elif fn in ("min", "max"):
                mins, maxs, valid = st.finalize()
                outs.append([int(s) if ok else None for s, ok in zip(mins if fn == "min" else maxs, valid)])
            elif fn in ("any", "all"):
                vals, valid = st.finalize()
                outs.append([bool(s) if ok else None for s, ok in zip(vals, valid)])
            else:
                G = grouper.num_groups
                # groups that never got a row of this aggregate's lane still exist: one null row each keeps them, in id order
                rows = [(g, None) for g in range(G)] + st["rows"]
                gid = from_list([g for g, _ in rows], np.int32)
                v = from_list([x for _, x in rows], np.int64)
                w = O.groupby_mean_i64(np.ascontiguousarray(gid.values), None, 0, np.ascontiguousarray(v.values), v.valid_bitmap(), 0,
                                       len(rows), skip_nulls=True, min_count=st["minc"])
                had_null = np.zeros(G, bool)
                for g, x in st["rows"]:
                    had_null[g] |= x is None
                assert w["keys"].tolist() == list(range(G))
                outs.append([float(m) if ok and (st["skip"] or not had_null[g]) else None
                             for g, (m, ok) in enumerate(zip(w["means"], w["valid"]))])
        return keys, outs
    return run


@pytest.mark.parametrize("two_states", [False, True], ids=["serial", "merged"])
def test_golden_grouped_aggregates(two_states):
    """acero/hash_aggregate_test.cc — CountOnly :714, MeanOnly :959, MeanOverflow :1048, MinMaxOnly :1591, MinMaxTypes :1661,
    AnyAndAll :2071, AnyAllSlicedNullableBoolean :2160, CountAndSum :3293, SumMeanProductKeepNulls :3481 — replayed on the
    oracle's kernel states (HashCountState / HashSumState / HashMinMaxState / HashBoolState, groupby_mean_i64) over
    the oracle Grouper's ids, one state and two merged states."""
    pytest.importorskip("pyarrow")
    from . import golden_hash_aggregate as H

    assert H.replay(GOLD, _oracle_group_by(two_states), key_types=(pa.int64(), pa.int32())) == 56


@pytest.mark.parametrize("use_threads", [False, True])
def test_golden_grouped_aggregates_transcription_holds_on_the_reference_build(use_threads):
    """The same vectors through the stock pyarrow 25.0.0 build (Table.group_by and the "aggregate" exec node), so that a
    transcription slip (the x8 restatement on int64, the hash_min_max -> hash_min + hash_max split) cannot hide."""
    pytest.importorskip("pyarrow")
    from . import golden_hash_aggregate as H

    assert H.replay(GOLD, H.stock_group_by(use_threads), key_types=(pa.int64(), pa.int32())) == 56
    assert H.replay(GOLD, H.declaration_group_by("aggregate", use_threads)) == 28


# ------------------------------------------------------------------ comparisons and arithmetic: the reference's known answers
SCALAR_GOLD = json.load(open(os.path.join(HERE, "golden", "reference_vectors_scalar.json")))


def _oracle_scalar_op(fn, left, right):
    """golden_scalar_ops' `run` on the oracle: O.compare / O.arith / O.divide see the values of every slot (what lies
    under a null included, as the kernels do) plus "both operands valid"; the result validity is the AND of the operand
    validities (PropagateNulls), a null scalar nulls everything; a failing slot raises the reference's Status text."""
    def split(x):
        if isinstance(x, pa.Scalar):
            return (x.as_py() if x.is_valid else 0), None if x.is_valid else False
        np_dtype = np.dtype(x.type.to_pandas_dtype()) if not pa.types.is_timestamp(x.type) else np.dtype(np.int64)
        phys = x.cast(pa.int64()) if pa.types.is_timestamp(x.type) else x
        vals = np.frombuffer(phys.buffers()[1], dtype=np_dtype, count=len(x) + x.offset)[x.offset:] if len(x) else np.zeros(0, np_dtype)
        return vals, np.array([v is not None for v in x.to_pylist()], dtype=bool)

    typ = left.type if isinstance(left, pa.Array) else right.type
    both_scalar = isinstance(left, pa.Scalar) and isinstance(right, pa.Scalar)
    n = 1 if both_scalar else len(left if isinstance(left, pa.Array) else right)
    (lv, lok), (rv, rok) = split(left), split(right)
    valid = np.ones(n, bool)
    for ok in (lok, rok):
        if ok is False:
            valid[:] = False
        elif ok is not None:
            valid &= ok
    np_dtype = np.dtype(np.int64) if pa.types.is_timestamp(typ) else np.dtype(typ.to_pandas_dtype())
    if both_scalar:
        lv, rv = np.array([lv], np_dtype), np.array([rv], np_dtype)
    if fn in ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal"):
        cast = lambda v: v if isinstance(v, np.ndarray) else np_dtype.type(v)
        data, out_type = O.compare(fn, cast(lv), cast(rv)), pa.bool_()
        data = np.broadcast_to(data, (n,))
    else:
        op, checked = (fn[:-8], True) if fn.endswith("_checked") else (fn, False)
        if op == "divide":
            data, error = O.divide(lv, rv, valid, checked, dtype=np_dtype)
            if error is not None:
                raise pa.ArrowInvalid(error)
        else:
            data, overflowed = O.arith(op, lv, rv, valid, dtype=np_dtype)
            if checked and overflowed:
                raise pa.ArrowInvalid("overflow")
        out_type = typ
    out = pa.array(np.asarray(data), type=out_type, mask=~valid)
    return out[0] if both_scalar else out


def test_golden_compare_and_arithmetic_on_the_oracle():
    """kernels/scalar_compare_test.cc:251-456 (SimpleCompareArrayScalar / ScalarArray / ArrayArray, TestNullScalar,
    TestCompareTimestamps.Basics) and kernels/scalar_arithmetic_test.cc:562-939 (Add / Sub / Mul / Div of the integral,
    signed, unsigned and floating fixtures: wrap-around, the checked forms' "overflow", overflow hidden under a null,
    "divide by zero", min / -1), every numeric type — replayed on O.compare / O.arith / O.divide."""
    pytest.importorskip("pyarrow")
    from . import golden_scalar_ops as S

    ran = 0
    for case in S.cases(SCALAR_GOLD):
        S.check(case, _oracle_scalar_op)
        ran += 1
    assert ran == 1982


def test_golden_compare_and_arithmetic_transcription_holds_on_the_reference_build():
    pytest.importorskip("pyarrow")
    from . import golden_scalar_ops as S

    ran = sum(1 for case in S.cases(SCALAR_GOLD) if S.check(case, lambda fn, l, r: pc.call_function(fn, [l, r])) is not None or True)
    assert ran == 1982


def _oracle_scalar_argument_batches(case, aggs):
    """golden_hash_aggregate.replay_scalar_arguments' `run` on the oracle's kernel states: a scalar batch is consumed
    through the states' `scalar=` form (the reference's ExecSpan with a scalar in slot 0)."""
    grouper = O.Grouper(1)
    states = []
    for fn, opts in aggs:
        skip = True if opts is None or fn == "count" else opts["skip_nulls"]
        minc = 1 if opts is None or fn == "count" else opts["min_count"]
        states.append(O.HashCountState(opts["mode"] if opts else "only_valid") if fn == "count" else
                      O.HashSumState(skip, minc) if fn in ("sum", "mean") else
                      O.HashMinMaxState(skip) if fn in ("min", "max") else O.HashBoolState(fn == "all", skip, minc))
    mean_counts = {}
    for b in case["batches"]:
        k = from_list(b["key"], np.int64)
        ids = grouper.consume([(np.ascontiguousarray(k.values), None)])
        for (fn, opts), st in zip(aggs, states):
            st.resize(grouper.num_groups)
            if "scalar" in b:
                v, ok = b["scalar"], b["scalar"] is not None
                if fn == "count":
                    st.consume(None, 0, ids, scalar_valid=ok)
                elif fn in ("any", "all"):
                    st.consume(None, None, 0, ids, scalar=(ok, bool(v)))
                else:
                    st.consume(None, None, 0, ids, scalar=(0 if v is None else v, ok))
            else:
                xs = b["argument"]
                if fn == "count":
                    a = from_list([None if x is None else 0 for x in xs], np.int64)
                    st.consume(a.valid_bitmap(), 0, ids)
                elif fn in ("any", "all"):
                    a = from_list(xs, np.bool_)
                    st.consume(np.asarray(a.values, bool), a.valid_bitmap(), 0, ids)
                else:
                    a = from_list(xs, np.int64)
                    st.consume(np.ascontiguousarray(a.values), a.valid_bitmap(), 0, ids)
    (kv, kvalid), = grouper.uniques([np.int64])
    outs = []
    for (fn, opts), st in zip(aggs, states):
        if fn == "count":
            outs.append(st.counts.tolist())
        elif fn in ("sum", "mean"):
            sums, valid, _ = st.finalize()
            # hash_mean = double(sum) / count while the sums are exact (the int32 values here are tiny)
            outs.append([(int(s) if fn == "sum" else float(s) / int(c)) if ok else None for s, c, ok in zip(sums, st.counts, valid)])
        elif fn in ("min", "max"):
            mins, maxs, valid = st.finalize()
            outs.append([int(x) if ok else None for x, ok in zip(mins if fn == "min" else maxs, valid)])
        else:
            vals, valid = st.finalize()
            outs.append([bool(x) if ok else None for x, ok in zip(vals, valid)])
    return [int(v) for v in kv], outs


def test_golden_grouped_aggregates_with_scalar_arguments():
    """acero/hash_aggregate_test.cc CountScalar :799, SumMeanProductScalar :1010, MinMaxScalar :1970, AnyAllScalar :2198 —
    on the oracle's kernel states (their scalar-broadcast consume) and on the stock wheel (union of projected literals)."""
    pytest.importorskip("pyarrow")
    from . import golden_hash_aggregate as H

    assert H.replay_scalar_arguments(GOLD, _oracle_scalar_argument_batches) == 3
    for threads in (False, True):
        assert H.replay_scalar_arguments(GOLD, H.union_of_scalar_batches("aggregate", threads)) == 3


def test_golden_numeric_casts_transcription_holds_on_the_reference_build():
    """kernels/scalar_cast_test.cc:269-431 as data (tests/golden/reference_vectors_scalar.json, cast_numeric): the 51
    cases hold on the stock wheel; the plugin replay (tests/test_gpu_arrow_plugin.py, GOLDEN_SCALAR_OPS_SCRIPT) holds the
    device route to the same values and the same error texts."""
    pytest.importorskip("pyarrow")
    from . import golden_scalar_ops as S

    results = [S.check_cast(c, lambda arr, to, **o: pc.cast(arr, options=pc.CastOptions(target_type=to, **o)))
               for c in S.cast_cases(SCALAR_GOLD)]
    assert len(results) == 51 and sum(isinstance(r, str) for r in results) == 20


def test_golden_kleene_logic_on_the_oracle_and_the_reference_build():
    """kernels/scalar_boolean_test.cc:54-152 (Invert, KleeneAnd, KleeneOr and their scalar forms): O.kleene and the stock
    wheel against the transcription."""
    pytest.importorskip("pyarrow")
    from . import golden_scalar_ops as S

    ran = 0
    for fn, args, want in S.boolean_cases(SCALAR_GOLD):
        assert pc.call_function(fn, args).equals(want), (fn, args)
        if fn != "invert":
            n = len(want)
            cols = []
            for x in args:
                xs = x.to_pylist() if isinstance(x, pa.Array) else [x.as_py()] * n
                cols.append((np.array([bool(v) for v in xs]), np.array([v is not None for v in xs])))
            data, valid = O.kleene("and" if fn == "and_kleene" else "or", cols[0][0], cols[0][1], cols[1][0], cols[1][1])
            assert [bool(d) if v else None for d, v in zip(data, valid)] == want.to_pylist(), (fn, args)
        ran += 1
    assert ran == 43


@pytest.mark.skipif(pa is None, reason="pyarrow (the reference build) is not installed")
@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.int8, np.uint16, np.float32, np.float64])
def test_hash_product_and_first_last_restatements_against_the_reference_build(dtype):
    """O.hash_product_row_order and O.group_edge_rows — the restatements the kernel tier trusts for hash_product / hash_first /
    hash_last / hash_one — against the reference's own GroupByNode on one thread (one state = row order): products bit for
    bit (wrapping integers, doubles in row order), first / last non-null values, skip_nulls on and off."""
    rng = np.random.default_rng(77 + np.dtype(dtype).itemsize + (np.dtype(dtype).kind == "f"))
    n, G = 5000, 37
    gids = rng.integers(0, G, n).astype(np.uint32)
    valid = rng.random(n) > 0.25
    valid[gids == 5] = False                                           # a group of nulls only
    if np.dtype(dtype).kind == "f":
        vals = (1.0 + rng.standard_normal(n) * 0.7).astype(dtype)
        vals[7], vals[11] = -0.0, dtype(1e30)
    else:
        info = np.iinfo(dtype)
        vals = rng.integers(max(info.min, -9), min(info.max, 11), n).astype(dtype)
        vals[rng.random(n) < 0.2] = dtype(info.max - 2) if info.max > 300 else dtype(3)
    t = pa.table({"g": pa.array(gids), "v": pa.array(vals, mask=~valid)})
    ref = t.group_by("g", use_threads=False).aggregate([("v", "product"), ("v", "first"), ("v", "last"), ("v", "one"),
                                                        ("v", "first", pc.ScalarAggregateOptions(skip_nulls=False)),
                                                        ("v", "last", pc.ScalarAggregateOptions(skip_nulls=False))]).sort_by("g")
    assert ref.num_rows == G
    prods, counts, _seen = O.hash_product_row_order(vals, valid, gids, G)
    want = ref.column("v_product")
    for g in range(G):
        w = want[g].as_py()
        if counts[g] == 0:
            assert w is None
        elif np.dtype(dtype).kind == "f":
            assert np.float64(w).view(np.uint64) == prods[g].view(np.uint64) or (np.isnan(w) and np.isnan(prods[g])), (g, w, prods[g])
        else:
            assert np.uint64(w % (1 << 64)) == prods[g], (g, w, prods[g])
    first_rows, first_has = O.group_edge_rows(gids, valid, G, last=False)
    last_rows, last_has = O.group_edge_rows(gids, valid, G, last=True)
    any_first, _ = O.group_edge_rows(gids, None, G, last=False)
    any_last, _ = O.group_edge_rows(gids, None, G, last=True)
    cols = [ref.column(i) for i in range(ref.num_columns)]
    names = ref.schema.names
    first_c, last_c, one_c = cols[names.index("v_first")], cols[names.index("v_last")], cols[names.index("v_one")]
    keep_first, keep_last = cols[-2], cols[-1]
    same = (lambda a, b: np.asarray([a], dtype=dtype).view(np.uint8).tobytes() == np.asarray([b], dtype=dtype).view(np.uint8).tobytes())
    for g in range(G):
        for col, rows, has in ((first_c, first_rows, first_has), (last_c, last_rows, last_has), (one_c, first_rows, first_has)):
            w = col[g].as_py()
            assert (w is None) == (not has[g]), (g, w)
            if has[g]:
                assert same(w, vals[rows[g]]), (g, w, vals[rows[g]])
        # skip_nulls = false: null where a null row comes before (after) the first (last) value
        for col, rows, has, edge in ((keep_first, first_rows, first_has, any_first), (keep_last, last_rows, last_has, any_last)):
            w = col[g].as_py()
            want_valid = bool(has[g]) and rows[g] == edge[g]
            assert (w is not None) == want_valid, (g, w, want_valid)
            if want_valid:
                assert same(w, vals[rows[g]])


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int64, np.int16])
def test_grouped_moments_restatement_against_the_reference_build(dtype):
    """O.grouped_moments / O.moments_merge / O.moments_statistic — the restatement of GroupedStatisticImpl the kernel tier
    compares the device moments with — against the reference's own GroupByNode on one thread over a table of two chunks
    (= two batches: two-pass moments per batch, Moments::Merge between them): bit for bit where the reference takes its generic
    path (floats, int64), to rounding for small integers (its ConsumeIntegral sums them exactly); nulls where Finalize leaves
    a group null."""
    rng = np.random.default_rng(99 + np.dtype(dtype).itemsize)
    n, G = 6000, 41
    gids = rng.integers(0, G, n).astype(np.uint32)
    gids[:9] = [G, G, G, G, G + 1, G + 1, G + 1, G + 2, G + 2]         # groups of 4, 3 and 2 values (unbiased skew / kurtosis, ddof)
    G += 3
    valid = rng.random(n) > 0.2
    valid[gids == 5] = False                                           # a group of nulls only
    valid[:9] = True
    vals = (1e4 + rng.standard_normal(n) * 30).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(-3000, 3000, n).astype(dtype)
    half = n // 2 + 7
    chunks = [pa.table({"g": pa.array(gids[a:b]), "v": pa.array(vals[a:b], mask=~valid[a:b])}) for a, b in ((0, half), (half, n))]
    t = pa.concat_tables(chunks)
    V, S = pc.VarianceOptions, pc.SkewOptions
    # (function, options, stat, ddof, biased, min_count, skip_nulls)
    cases = [("variance", V(ddof=0), 0, 0, True, 0, True), ("variance", V(ddof=2, min_count=4), 0, 2, True, 4, True),
             ("stddev", V(ddof=1, skip_nulls=False), 1, 1, True, 0, False), ("skew", S(), 2, 0, True, 0, True),
             ("skew", S(biased=False), 2, 0, False, 0, True), ("kurtosis", S(biased=False, min_count=5), 3, 0, False, 5, True),
             ("kurtosis", S(skip_nulls=False), 3, 0, True, 0, False)]
    ref = t.group_by("g", use_threads=False).aggregate([("v", f, o) for f, o, *_ in cases]).sort_by("g")
    assert ref.num_rows == G
    state = None
    for a, b in ((0, half), (half, n)):
        state = O.grouped_moments(vals[a:b], valid[a:b], gids[a:b], G, 4, state)
    moments, null_seen = state
    exact = np.dtype(dtype).itemsize == 8 or np.dtype(dtype).kind == "f"
    stat_cols = [i for i, name in enumerate(ref.schema.names) if name != "g"]      # (the key column comes first or last by version)
    assert len(stat_cols) == len(cases)
    for ci, (f, o, stat, ddof, biased, min_count, skip_nulls) in enumerate(cases):
        col = ref.column(stat_cols[ci])
        for g in range(G):
            w = col[g].as_py()
            mine = O.moments_statistic(moments[g], stat, ddof, biased)
            if mine is not None and (moments[g][0] < min_count or (not skip_nulls and null_seen[g])):
                mine = None
            assert (w is None) == (mine is None), (f, g, w, mine, moments[g])
            if w is None or (np.isnan(w) and np.isnan(mine)):
                continue
            if exact and stat <= 1:
                assert np.float64(w).view(np.uint64) == np.float64(mine).view(np.uint64), (f, g, w, mine)
            else:
                assert abs(w - mine) <= 1e-9 * max(1.0, abs(w)), (f, g, w, mine)


def test_grouped_moments_restatement_against_the_references_golden_vectors():
    """The known answers of the reference's own tests for the grouped moments (acero/hash_aggregate_test.cc:1080-1186:
    GroupBy.VarianceAndStddev with hash_skew / hash_kurtosis, and GroupBy.VarianceAndStddevDdof with ddof = 2) through
    O.grouped_moments / O.moments_statistic — the restatement the device kernels are compared with bit for bit."""
    values = np.array([1, 0, 0, 0, 4, 3, 0, -1, 1, 0], dtype=np.float64)
    valid = np.array([1, 0, 1, 0, 1, 1, 1, 1, 1, 0], dtype=bool)
    # keys 1, 1, 2, 3, null, 1, 2, 2, null, 3 -> dense ids in order of first appearance: 1 -> 0, 2 -> 1, 3 -> 2, null -> 3
    gids = np.array([0, 0, 1, 2, 3, 0, 1, 1, 3, 2], dtype=np.uint32)
    for dtype in (np.float64, np.int32, np.int64):       # (the reference runs the table for float64 and int32)
        moments, _ = O.grouped_moments(values.astype(dtype), valid, gids, 4, 4)
        golden = {  # group: (variance, stddev, skew, kurtosis) with the default options
            0: (1.0, 1.0, 0.0, -2.0), 1: (0.22222222222222224, 0.4714045207910317, -0.7071067811865478, -1.5), 2: None, 3: (2.25, 1.5, 0.0, -2.0)}
        for g, want in golden.items():
            got = [O.moments_statistic(moments[g], stat, 0, True) for stat in range(4)]
            if want is None:
                assert got == [None] * 4, (g, got)
            else:
                assert np.allclose(got, want, rtol=1e-12, atol=1e-15), (g, got, want)
        ddof2 = {0: None, 1: (0.6666666666666667, 0.816496580927726), 2: None, 3: None}
        for g, want in ddof2.items():
            got = [O.moments_statistic(moments[g], stat, 2, True) for stat in (0, 1)]
            if want is None:
                assert got == [None, None], (g, got)
            else:
                assert np.allclose(got, want, rtol=1e-12), (g, got, want)


def test_oracle_rank_against_the_references_known_answers_and_pyarrow():
    """Round 6: the oracle's restatement of vector_rank.cc — pinned on the known answers of the reference's own TestRank
    (tests/golden/rank_vectors.json: vector_sort_test.cc:2408-2507, floats and integers) and, where pyarrow is importable,
    against the reference build on seeded arrays (every tiebreaker, order, null placement; quantile ranks bit for bit)."""
    from . import parity_cases as P

    cases = P.rank_golden_cases()
    assert len(cases) == 200
    for v, valid, order, place, tb, expected in cases:
        got = O.rank(v, valid, order == "descending", place == "at_start", tb)
        assert got.tolist() == expected, (v.dtype, order, place, tb, got.tolist(), expected)
    pa = pytest.importorskip("pyarrow")
    import warnings

    import pyarrow.compute as pc

    rng = np.random.default_rng(20260930)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)      # (null_placement in RankOptions: deprecated in 25.0, same result)
        for dt in (np.int64, np.uint32, np.float64, np.float32):
            for n in (0, 1, 7, 600):
                for null_p in (0.0, 0.3):
                    if np.dtype(dt).kind == "f":
                        v = rng.integers(-5, 5, n).astype(dt)
                        v[rng.random(n) < 0.2] = np.nan
                        v[rng.random(n) < 0.1] = -0.0
                    else:
                        v = rng.integers(0, 9, n).astype(dt)
                    valid = rng.random(n) >= null_p if null_p else None
                    a = pa.array(v, mask=None if valid is None else ~valid)
                    for order in ("ascending", "descending"):
                        for place in ("at_end", "at_start"):
                            for tb in ("first", "min", "max", "dense"):
                                w = pc.rank(a, sort_keys=order, null_placement=place, tiebreaker=tb).to_numpy()
                                assert (w == O.rank(v, valid, order == "descending", place == "at_start", tb)).all(), (dt, n, order, place, tb)
                            w = pc.rank_quantile(a, sort_keys=order, null_placement=place).to_numpy()
                            g = O.rank(v, valid, order == "descending", place == "at_start", "quantile")
                            assert (w.view(np.uint64) == g.view(np.uint64)).all(), (dt, n, order, place, "quantile")
                            w = pc.rank_normal(a, sort_keys=order, null_placement=place).to_numpy()
                            g = O.rank(v, valid, order == "descending", place == "at_start", "normal")
                            assert P.max_ulps(w, g) <= P.RANK_NORMAL_ULPS, (dt, n, order, place, "normal")


# NormalPPF's known answers in the reference: cpp/src/arrow/util/math_test.cc:28-82 (Scipy's norm.ppf and Wichura's paper;
# EXPECT_DOUBLE_EQ = within 4 ULPs), and the rank_normal answers of TestRankQuantile, kernels/vector_sort_test.cc:2726-2790
# (compared with atol 1e-8 there, :2656-2660).
PPF_VECTORS = [(0.0, -np.inf), (0.001, -3.090232306167813), (0.01, -2.3263478740408408), (0.02, -2.053748910631823),
               (0.03, -1.880793608151251), (0.04, -1.75068607125217), (0.05, -1.6448536269514729), (0.06, -1.5547735945968535),
               (0.07, -1.4757910281791706), (0.08, -1.4050715603096329), (0.09, -1.3407550336902165), (0.1, -1.2815515655446004),
               (0.2, -0.8416212335729142), (0.3, -0.5244005127080409), (0.4, -0.2533471031357997), (0.5, 0.0),
               (0.6, 0.2533471031357997), (0.7, 0.5244005127080407), (0.8, 0.8416212335729143), (0.9, 1.2815515655446004),
               (0.91, 1.3407550336902165), (0.92, 1.4050715603096329), (0.93, 1.475791028179171), (0.94, 1.5547735945968535),
               (0.95, 1.6448536269514722), (0.96, 1.7506860712521692), (0.97, 1.8807936081512509), (0.98, 2.0537489106318225),
               (0.99, 2.3263478740408408), (0.999, 3.090232306167813), (1.0, np.inf),
               (0.25, -0.6744897501960817), (0.001, -3.090232306167814), (1e-20, -9.262340089798408)]
RANK_NORMAL_VECTORS = [   # (values, valid, order, null placement, expected)
    ([1, 2, 1, 2, 1], None, "ascending", "at_end", [-0.5244005127080409, 0.8416212335729143, -0.5244005127080409, 0.8416212335729143, -0.5244005127080409]),
    ([1, 2, 1, 2, 1], None, "descending", "at_start", [0.5244005127080407, -0.8416212335729142, 0.5244005127080407, -0.8416212335729142, 0.5244005127080407]),
    ([0, 1, 0, 2, 0], [0, 1, 0, 1, 0], "ascending", "at_start", [-0.5244005127080409, 0.5244005127080407, -0.5244005127080409, 1.2815515655446004, -0.5244005127080409]),
    ([0, 1, 0, 2, 0], [0, 1, 0, 1, 0], "ascending", "at_end", [0.5244005127080407, -1.2815515655446004, 0.5244005127080407, -0.5244005127080409, 0.5244005127080407]),
    ([0, 1, 0, 2, 0], [0, 1, 0, 1, 0], "descending", "at_start", [-0.5244005127080409, 1.2815515655446004, -0.5244005127080409, 0.5244005127080407, -0.5244005127080409]),
    ([0, 1, 0, 2, 0], [0, 1, 0, 1, 0], "descending", "at_end", [0.5244005127080407, -0.5244005127080409, 0.5244005127080407, -1.2815515655446004, 0.5244005127080407]),
    ([7, 5, 5, 4, 4, 3, 3, 3, 2, 1], None, "ascending", "at_end",
     [1.6448536269514722, 0.8416212335729143, 0.8416212335729143, 0.2533471031357997, 0.2533471031357997, -0.38532046640756773,
      -0.38532046640756773, -0.38532046640756773, -1.0364333894937898, -1.6448536269514729]),
    ([7, 5, 5, 4, 4, 3, 3, 3, 2, 1], None, "descending", "at_start",
     [-1.6448536269514729, -0.8416212335729142, -0.8416212335729142, -0.2533471031357997, -0.2533471031357997, 0.38532046640756773,
      0.38532046640756773, 0.38532046640756773, 1.0364333894937898, 1.6448536269514722]),
    ([0], [0], "ascending", "at_end", [0.0]),
]


def test_oracle_normal_ppf_and_rank_normal_against_the_references_known_answers():
    from . import parity_cases as P

    for p, want in PPF_VECTORS:
        got = float(O.normal_ppf(np.array([p]))[0])
        assert P.max_ulps([got], [want]) <= 4, (p, got, want)
    for v, valid, order, place, want in RANK_NORMAL_VECTORS:
        got = O.rank(np.array(v, np.int64), None if valid is None else np.array(valid, bool), order == "descending", place == "at_start", "normal")
        assert np.allclose(got, want, rtol=0, atol=1e-8) and P.max_ulps(got, want) <= 4, (v, order, place, got)
