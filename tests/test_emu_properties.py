"""Property-style parity on the CPU emulator: hypothesis draws the shape parameters (lengths around
word / tile boundaries, bit offsets, null and selection densities, index types), the seeded generators
of tests/util.py build the arrays, and the checks of tests/parity_cases.py compare the kernel sources
with the C / numpy oracle bit for bit.  Complements the fixed grids of test_emu_parity.py."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from . import parity_cases as P
from . import util as U

pytestmark = pytest.mark.emu

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow],
              derandomize=True)
lengths = st.one_of(st.integers(0, 200), st.sampled_from([63, 64, 65, 127, 128, 129, 4095, 4096, 4097]),
                    st.integers(200, 9000))
offsets = st.integers(0, 130)
density = st.sampled_from([0.0, 0.01, 0.1, 0.5, 0.9, 1.0])
nulls = st.sampled_from([0.0, 0.0, 0.05, 0.5, 1.0])


@settings(max_examples=200, **COMMON)
@given(n=lengths, voff=offsets, moff=offsets, true_p=density, vnull=nulls, mnull=nulls,
       dtype=st.sampled_from([np.int8, np.int16, np.int32, np.int64, np.float64]),
       sel=st.sampled_from(["drop", "emit_null"]), seed=st.integers(0, 2**31 - 1))
def test_filter_property(emu_ctx, n, voff, moff, true_p, vnull, mnull, dtype, sel, seed):
    rng = np.random.default_rng(seed)
    v = U.random_array(rng, dtype, n, null_p=vnull, offset=voff, tail=3)
    m = U.random_mask(rng, n, true_p, null_p=mnull, offset=moff, tail=5)
    P.check_filter(emu_ctx, v, m, sel, use_pyarrow=False)
    P.check_mask_to_indices(emu_ctx, m, sel)


@settings(max_examples=150, **COMMON)
@given(nv=st.integers(1, 5000), m=lengths, voff=offsets, ioff=offsets, vnull=nulls, inull=nulls,
       dtype=st.sampled_from([np.uint8, np.int32, np.int64, np.float32]),
       idx_dtype=st.sampled_from([np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]),
       seed=st.integers(0, 2**31 - 1))
def test_take_property(emu_ctx, nv, m, voff, ioff, vnull, inull, dtype, idx_dtype, seed):
    rng = np.random.default_rng(seed)
    nv = min(nv, np.iinfo(idx_dtype).max)
    v = U.random_array(rng, dtype, nv, null_p=vnull, offset=voff, tail=2)
    i = U.random_array(rng, idx_dtype, m, null_p=inull, offset=ioff, tail=1, lo=0, hi=nv - 1)
    P.check_take(emu_ctx, v, i, use_pyarrow=False)


@settings(max_examples=80, **COMMON)
@given(n=lengths, loff=offsets, roff=offsets, lnull=nulls, rnull=nulls, seed=st.integers(0, 2**31 - 1))
def test_kleene_property(emu_ctx, n, loff, roff, lnull, rnull, seed):
    rng = np.random.default_rng(seed)
    left = U.random_mask(rng, n, 0.5, null_p=lnull, offset=loff, tail=3)
    right = U.random_mask(rng, n, 0.5, null_p=rnull, offset=roff, tail=1)
    P.check_kleene_and_invert(emu_ctx, left, right, use_pyarrow=False)


@settings(max_examples=80, **COMMON)
@given(nv=st.integers(1, 800), m=st.integers(0, 6000), vnull=nulls, inull=nulls, max_len=st.sampled_from([0, 1, 7, 40, 300]),
       voff=st.integers(0, 70), seed=st.integers(0, 2**31 - 1))
def test_binary_take_property(emu_ctx, nv, m, vnull, inull, max_len, voff, seed):
    rng = np.random.default_rng(seed)
    v = U.random_binary(rng, nv, null_p=vnull, offset=voff, tail=2, max_len=max_len)
    i = U.random_array(rng, np.int32, m, null_p=inull, offset=1, lo=0, hi=nv - 1)
    P.check_binary_take(emu_ctx, v, i, use_pyarrow=False)


@settings(max_examples=60, **COMMON)
@given(n=st.integers(0, 5000), groups=st.sampled_from([1, 3, 50, 2000]), knull=nulls, vnull=nulls,
       skip_nulls=st.booleans(), min_count=st.sampled_from([0, 1, 3]), batches=st.integers(1, 3),
       seed=st.integers(0, 2**31 - 1))
def test_groupby_property(emu_ctx, n, groups, knull, vnull, skip_nulls, min_count, batches, seed):
    rng = np.random.default_rng(seed)
    k = U.random_array(rng, np.int32, n, null_p=knull, offset=2, lo=-groups, hi=groups)
    v = U.random_array(rng, np.int64, n, null_p=vnull, offset=1)
    P.check_groupby_sum(emu_ctx, k, v, skip_nulls, min_count, batches=batches, use_pyarrow=False)
    P.check_groupby_min_max(emu_ctx, k, v, skip_nulls, batches=batches, use_pyarrow=False)


@settings(max_examples=120, **COMMON)
@given(kind=st.sampled_from(["int16", "int64", "bool", "utf8"]),
       specs=st.lists(st.tuples(st.one_of(st.integers(0, 70), st.sampled_from([63, 64, 65, 128, 500])), nulls,
                                st.integers(0, 130)), min_size=1, max_size=6),
       seed=st.integers(0, 2**31 - 1))
def test_concat_arrays_property(emu_ctx, kind, specs, seed):
    """Concatenate: any number of chunks of any length (also empty) glued at any bit position, with and without
    validity; the numpy concatenation of the logical rows is the oracle."""
    rng = np.random.default_rng(seed)
    if kind == "bool":
        chunks = [U.random_mask(rng, n, 0.5, null_p=p, offset=o, tail=1) for n, p, o in specs]
    elif kind == "utf8":
        chunks = [U.random_binary(rng, n, null_p=p, offset=o, tail=1, utf8=True, max_len=9) for n, p, o in specs]
    else:
        chunks = [U.random_array(rng, np.dtype(kind).type, n, null_p=p, offset=o, tail=1) for n, p, o in specs]
    P.check_concat_arrays(emu_ctx, chunks, use_pyarrow=False)


@settings(max_examples=150, **COMMON)
@given(n=st.one_of(st.integers(1, 300), st.sampled_from([4095, 4096, 4097, 8193])),
       shape=st.sampled_from([(128, 4), (128, 1), (256, 8), (256, 2), (512, 4)]),
       spread=st.sampled_from([0, 1, 7, 1000, 2**31, 2**62]), byte_width=st.sampled_from([4, 8]),
       seed=st.integers(0, 2**31 - 1))
def test_delta_binary_packed_property(emu_ctx, n, shape, spread, byte_width, seed):
    """DELTA_BINARY_PACKED: any legal block shape, any delta magnitude (0 = constant column, 2**62 = wrap-around),
    counts around miniblock / block / scan-tile boundaries; the restated decoder is the oracle, the header walk must
    agree with it on every field."""
    from oracle import oracle as O

    rng = np.random.default_rng(seed)
    deltas = rng.integers(-spread, spread, n, endpoint=True) if spread else np.zeros(n, dtype=np.int64)
    values = np.cumsum(deltas.astype(np.uint64)).astype(np.int64) + rng.integers(-2**40, 2**40)   # (wraps modulo 2**64)
    page = O.delta_binary_packed_encode(values, *shape)
    want, used = O.delta_binary_packed_decode(page)
    assert used == len(page) and np.array_equal(want, values)
    mbs, vpm, total, first, consumed = emu_ctx.parquet.scan_delta_miniblocks(page + b"\x01\x02")
    assert (vpm, total, first, consumed) == (shape[0] // shape[1], n, int(values[0]), len(page))
    out = emu_ctx.parquet.decode_delta_binary_packed(page, byte_width)
    got = out.data.cpu().numpy()[: n * byte_width].view(np.int64 if byte_width == 8 else np.int32)
    assert np.array_equal(got, values if byte_width == 8 else values.astype(np.int32))


@settings(max_examples=120, **COMMON)
@given(n=lengths, loff=offsets, roff=offsets, lnull=nulls, rnull=nulls, zero_p=st.sampled_from([0.0, 0.0, 0.01, 0.5]),
       kind=st.sampled_from(["i", "f"]), checked=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_divide_property(emu_ctx, n, loff, roff, lnull, rnull, zero_p, kind, checked, seed):
    """divide / divide_checked: any operand offsets, null densities and zero-divisor densities; either the oracle's error
    (named by the last failing valid slot) or its values at every visited slot."""
    from oracle import oracle as O

    rng = np.random.default_rng(seed)
    if kind == "i":
        l = U.random_array(rng, np.int64, n, null_p=lnull, offset=loff, tail=2)
        r = U.random_array(rng, np.int64, n, null_p=rnull, offset=roff, tail=3, lo=-5, hi=5)
        if n > 3:
            l.values[loff + 1], r.values[roff + 1] = -2**63, -1
    else:
        l = U.random_array(rng, np.float64, n, null_p=lnull, offset=loff, tail=2)
        r = U.random_array(rng, np.float64, n, null_p=rnull, offset=roff, tail=3)
        r.values[:] = np.round(r.values * 2) / 2
    r.values[rng.random(len(r.values)) < zero_p] = 0
    if zero_p == 0.0:
        r.values[r.values == 0] = 1
    both = l.logical_valid() & r.logical_valid()
    want, error = O.divide(l.logical_values(), r.logical_values(), both, checked)
    fn = emu_ctx.compute.divide_checked if checked else emu_ctx.compute.divide
    if error is not None:
        with pytest.raises(emu_ctx.ArrowInvalid) as e:
            fn(l.to_device(emu_ctx), r.to_device(emu_ctx))
        assert str(e.value) == error
        return
    out = fn(l.to_device(emu_ctx), r.to_device(emu_ctx))
    got = out.to_numpy()[0]
    ok = both & ~(np.isnan(want) if kind == "f" else np.zeros(n, bool))
    assert np.array_equal(got[ok], want[ok])
    if kind == "f":
        assert np.array_equal(np.isnan(got[both]), np.isnan(want[both]))
