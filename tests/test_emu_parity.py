"""Kernel-logic parity on the CPU: the *unchanged* kernel sources of arrow_amd/csrc compiled for
the host against tests/emu/hip_emu.h (a fiber-based SIMT emulator) and driven through the same
arrow_amd.compute -> C ABI path as on the GPU.  These are not GPU compute calls and not a CPU
fallback of the product: the emulated library is built and loaded by the tests only.
Sizes are small (the emulator context-switches at every wave collective)."""
import numpy as np
import pytest

from oracle import oracle as O

from . import parity_cases as P
from . import util as U

pytestmark = pytest.mark.emu


def rng_for(*key):
    return np.random.default_rng([U.kRandomSeed, *[abs(hash(k)) % (1 << 31) for k in key]])


# ------------------------------------------------------------------ filter
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 129, 4095, 4096, 4097, 9000])
def test_filter_int64_lengths(emu_ctx, n, sel):
    rng = rng_for("f64len", n, sel)
    v = U.random_array(rng, np.int64, n, null_p=0.1)
    m = U.random_mask(rng, n, 0.3, null_p=0.05)
    P.check_filter(emu_ctx, v, m, sel)


@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("true_p", [0.0, 0.1, 0.5, 0.999, 1.0])
@pytest.mark.parametrize("vnull,mnull", [(0.0, 0.0), (0.1, 0.0), (0.0, 0.05), (0.999, 0.5), (1.0, 1.0)])
def test_filter_int64_probabilities(emu_ctx, true_p, vnull, mnull, sel):
    """The grid of FilterRandomTest (vector_selection_test.cc:2241-2260), n scaled for the emulator."""
    rng = rng_for("fprob", true_p, vnull, mnull, sel)
    n = 5000
    v = U.random_array(rng, np.int64, n, null_p=vnull)
    m = U.random_mask(rng, n, true_p, null_p=mnull)
    P.check_filter(emu_ctx, v, m, sel)


@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("voff,moff", [(1, 0), (0, 3), (7, 13), (64, 65), (3, 4099)])
def test_filter_offsets(emu_ctx, voff, moff, sel):
    """Sliced inputs with non-zero, non-byte-aligned offsets (vector_selection_test.cc:286-302)."""
    rng = rng_for("foff", voff, moff, sel)
    n = 6000
    v = U.random_array(rng, np.int64, n, null_p=0.2, offset=voff, tail=5)
    m = U.random_mask(rng, n, 0.4, null_p=0.1, offset=moff, tail=9)
    P.check_filter(emu_ctx, v, m, sel)


@pytest.mark.parametrize("dtype", [np.int8, np.uint16, np.int32, np.float32, np.float64, np.uint64])
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
def test_filter_widths(emu_ctx, dtype, sel):
    rng = rng_for("fw", str(dtype), sel)
    n = 8200
    v = U.random_array(rng, dtype, n, null_p=0.1, offset=3)
    m = U.random_mask(rng, n, 0.5, null_p=0.05, offset=1)
    P.check_filter(emu_ctx, v, m, sel, use_pyarrow=True)


def test_filter_dense_and_batch_variants(emu_ctx):
    """The tuning knobs must never change results."""
    lib = emu_ctx._lib.get_lib()
    rng = rng_for("fvariants")
    v = U.random_array(rng, np.int64, 9000, null_p=0.1, offset=2)
    m = U.random_mask(rng, 9000, 0.1, null_p=0.05)
    try:
        for batch in (1, 4):
            for pipe in (0, 1):
                assert lib.arx_set_option(b"filter_batch", batch) == 0
                assert lib.arx_set_option(b"filter_pipe", pipe) == 0
                for sel in ("drop", "emit_null"):
                    P.check_filter(emu_ctx, v, m, sel, use_pyarrow=False)
    finally:
        lib.arx_set_option(b"filter_batch", 4)
        lib.arx_set_option(b"filter_pipe", 1)


def test_filter_no_nulls_has_no_validity(emu_ctx):
    rng = rng_for("fnn")
    v = U.random_array(rng, np.int64, 5000)
    m = U.random_mask(rng, 5000, 0.5)
    out = P.check_filter(emu_ctx, v, m, "drop")
    assert out.validity is None and out.null_count == 0


def test_filter_length_mismatch_is_invalid(emu_ctx):
    """vector_selection_test.cc:338-341."""
    v = U.HostArray(np.array([7, 8, 9], dtype=np.int64), None, 0, 3).to_device(emu_ctx)
    m = U.HostArray(np.zeros(0, dtype=bool), None, 0, 0).to_device(emu_ctx)
    for sel in ("drop", "emit_null"):
        with pytest.raises(emu_ctx.ArrowInvalid):
            emu_ctx.compute.filter(v, m, sel)


# ------------------------------------------------------------------ GetTakeIndices
@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("n,off", [(0, 0), (1, 0), (100, 5), (4097, 0), (9000, 3), (70000, 1)])
def test_mask_to_indices(emu_ctx, n, off, sel):
    rng = rng_for("m2i", n, off, sel)
    m = U.random_mask(rng, n, 0.2 if n < 20000 else 0.02, null_p=0.05, offset=off, tail=3)
    P.check_mask_to_indices(emu_ctx, m, sel)


def test_record_batch_filter_is_take_of_indices(emu_ctx):
    """FilterRecordBatch (vector_selection_filter_internal.cc:925-960): filter(v, m) must equal
    take(v, GetTakeIndices(m)) — the identity the reference's ValidateFilter relies on (:345-373)."""
    amd = emu_ctx
    rng = rng_for("rb")
    n = 5000
    a = U.random_array(rng, np.int64, n, null_p=0.1)
    b = U.random_array(rng, np.int32, n, null_p=0.0)
    m = U.random_mask(rng, n, 0.3, null_p=0.1)
    for sel in ("drop", "emit_null"):
        rb = amd.compute.RecordBatch({"a": a.to_device(amd), "b": b.to_device(amd)})
        out = amd.compute.filter(rb, m.to_device(amd), sel)
        for name, col in (("a", a), ("b", b)):
            direct = amd.compute.filter(col.to_device(amd), m.to_device(amd), sel)
            assert out.columns[name].to_pylist() == direct.to_pylist()


# ------------------------------------------------------------------ take
@pytest.mark.parametrize("idx_dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32,
                                       np.uint64, np.int64])
def test_take_index_types(emu_ctx, idx_dtype):
    """TakeRandomTest shape (values 1025, indices 257; vector_selection_test.cc:2282-2315)."""
    rng = rng_for("tk", str(idx_dtype))
    nv = 100 if np.dtype(idx_dtype).itemsize == 1 else 1025
    v = U.random_array(rng, np.int64, nv, null_p=0.1, offset=3)
    i = U.random_array(rng, idx_dtype, 257, null_p=0.1, offset=5, lo=0, hi=nv - 1)
    P.check_take(emu_ctx, v, i)


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.float32, np.float64])
@pytest.mark.parametrize("vnull,inull", [(0.0, 0.0), (0.1, 0.0), (0.0, 0.3), (1.0, 0.0), (0.5, 1.0)])
def test_take_widths_and_nulls(emu_ctx, dtype, vnull, inull):
    rng = rng_for("tkw", str(dtype), vnull, inull)
    v = U.random_array(rng, dtype, 3000, null_p=vnull)
    i = U.random_array(rng, np.int32, 1500, null_p=inull, lo=0, hi=2999)
    P.check_take(emu_ctx, v, i)


def test_take_empty_and_no_boundscheck(emu_ctx):
    rng = rng_for("tke")
    v = U.random_array(rng, np.int64, 50)
    P.check_take(emu_ctx, v, U.HostArray(np.zeros(0, dtype=np.int32), None, 0, 0))
    i = U.random_array(rng, np.uint32, 700, lo=0, hi=49)
    P.check_take(emu_ctx, v, i, boundscheck=False)


@pytest.mark.parametrize("bad", [9, -1])
def test_take_out_of_bounds(emu_ctx, bad):
    """IndexError naming the first offender (vector_selection_test.cc:1545-1556; int_util.cc:554)."""
    v = U.HostArray(np.arange(5, dtype=np.int64), None, 0, 5)
    idx = np.array([0, bad, 0, 77], dtype=np.int64)
    P.check_take_out_of_bounds(emu_ctx, v, U.HostArray(idx, None, 0, 4))


def test_take_null_index_is_not_bounds_checked(emu_ctx):
    v = U.HostArray(np.arange(5, dtype=np.int64), None, 0, 5)
    idx = U.HostArray(np.array([1, 99, 2], dtype=np.int32), np.array([True, False, True]), 0, 3)
    out = P.check_take(emu_ctx, v, idx)
    assert out.to_pylist() == [1, None, 2]


# ------------------------------------------------------------------ cast / compare / add
def _cast_inputs(rng, n):
    x = rng.standard_normal(n)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e39, -1e39, 3.4028235677973366e38,
                        1e-40, -1e-46, 1.0000000596046448, 1.00000017881393433, 2.0 ** -126, 2.0 ** -150])
    x[: len(special)] = special[: min(len(special), n)]
    big = rng.integers(0, n, size=n // 20)
    x[big] *= 1e40
    tiny = rng.integers(0, n, size=n // 20)
    x[tiny] *= 1e-42
    return x


@pytest.mark.parametrize("n,off", [(0, 0), (1, 0), (13, 1), (2048, 0), (2049, 3), (5000, 2)])
def test_cast_f64_f32(emu_ctx, n, off):
    rng = rng_for("cast", n, off)
    x = _cast_inputs(rng, n + off + 2) if n else np.zeros(off + 2)
    valid = rng.random(len(x)) > 0.1 if n % 2 else None
    P.check_cast_f64_f32(emu_ctx, U.HostArray(x, valid, off, n))


@pytest.mark.parametrize("n,off", [(1, 0), (64, 0), (127, 1), (513, 0), (3000, 5)])
def test_greater_f64_array_array(emu_ctx, n, off):
    rng = rng_for("gt", n, off)
    a = U.random_array(rng, np.float64, n, null_p=0.1, offset=off)
    b = U.random_array(rng, np.float64, n, null_p=0.1 if n % 2 else 0.0, offset=2 * off)
    a.values[rng.integers(0, len(a.values), 5)] = np.nan
    b.values[::7] = a.values[: len(b.values)][::7] if len(a.values) >= len(b.values) else 0.0
    P.check_greater_f64(emu_ctx, a, b)


def test_greater_scalar_forms_and_i64(emu_ctx):
    rng = rng_for("gts")
    a = U.random_array(rng, np.float64, 1000, null_p=0.1, offset=1)
    P.check_greater_f64(emu_ctx, a, 0.25)
    P.check_greater_f64(emu_ctx, -0.5, a)
    P.check_greater_f64(emu_ctx, a, float("nan"))
    x = U.random_array(rng, np.int64, 777, lo=-5, hi=5)
    y = U.random_array(rng, np.int64, 777, lo=-5, hi=5)
    P.check_greater_f64(emu_ctx, x, y)


def test_add(emu_ctx):
    rng = rng_for("add")
    x = U.random_array(rng, np.int64, 3001, null_p=0.1, offset=1)   # full range: wraps
    y = U.random_array(rng, np.int64, 3001, offset=3)
    P.check_add(emu_ctx, x, y)
    a = U.random_array(rng, np.float64, 1000, null_p=0.1)
    b = U.random_array(rng, np.float64, 1000, null_p=0.1)
    P.check_add(emu_ctx, a, b)


# ------------------------------------------------------------------ sort
@pytest.mark.parametrize("order", ["ascending", "descending"])
@pytest.mark.parametrize("placement", ["at_end", "at_start"])
def test_sort_small_range_with_ties_and_nulls(emu_ctx, order, placement):
    """Stability on ties + null placement (vector_sort_test.cc:640-960)."""
    rng = rng_for("sort1", order, placement)
    a = U.random_array(rng, np.uint64, 700, null_p=0.2, offset=3, lo=0, hi=7)
    P.check_sort_indices(emu_ctx, a, order, placement)


@pytest.mark.parametrize("dtype", [np.uint64, np.int64])
@pytest.mark.parametrize("n", [1, 2, 255, 4096, 4097, 9000])
def test_sort_full_range(emu_ctx, dtype, n):
    rng = rng_for("sort2", str(dtype), n)
    a = U.random_array(rng, dtype, n)
    P.check_sort_indices(emu_ctx, a, "ascending" if n % 2 else "descending", "at_end")


def test_sort_all_null_and_empty(emu_ctx):
    a = U.HostArray(np.arange(10, dtype=np.uint64), np.zeros(10, dtype=bool), 0, 10)
    P.check_sort_indices(emu_ctx, a)
    P.check_sort_indices(emu_ctx, U.HostArray(np.zeros(0, dtype=np.uint64), None, 0, 0))


# ------------------------------------------------------------------ group-by
@pytest.mark.parametrize("skip_nulls,min_count", [(True, 1), (False, 1), (True, 0), (True, 3), (False, 0)])
def test_groupby_sum_options(emu_ctx, skip_nulls, min_count):
    """min_count / keep-nulls semantics (acero/hash_aggregate_test.cc:3481-3530)."""
    rng = rng_for("gb", skip_nulls, min_count)
    k = U.random_array(rng, np.int32, 3000, null_p=0.05, offset=1, lo=-20, hi=20)
    v = U.random_array(rng, np.int64, 3000, null_p=0.2, offset=2)
    P.check_groupby_sum(emu_ctx, k, v, skip_nulls, min_count, batches=3)


def test_groupby_sum_wraparound_many_groups(emu_ctx):
    rng = rng_for("gbwrap")
    k = U.random_array(rng, np.int32, 5000, lo=0, hi=1500)
    v = U.random_array(rng, np.int64, 5000)  # full int64 range: sums wrap (to_unsigned add)
    P.check_groupby_sum(emu_ctx, k, v)


def test_groupby_sum_golden_sum_only(emu_ctx):
    """SumOnly (acero/hash_aggregate_test.cc:839-883) restated on int64 values: null key is its own
    group; a group whose values are all null sums to null."""
    keys = [1, 1, 2, 3, None, 1, 2, 2, None, 3]
    vals = [10, None, None, 1, 30, 5, None, 20, 40, None]
    k = U.HostArray(np.array([0 if x is None else x for x in keys], dtype=np.int32),
                    np.array([x is not None for x in keys]), 0, len(keys))
    v = U.HostArray(np.array([0 if x is None else x for x in vals], dtype=np.int64),
                    np.array([x is not None for x in vals]), 0, len(vals))
    got = P.check_groupby_sum(emu_ctx, k, v)
    assert got == [(0, 1, 15), (0, 2, 20), (0, 3, 1), (1, 0, 70)]


def test_hash_sum_direct_call_is_rejected(emu_ctx):
    """acero/hash_aggregate_test.cc:646-665 / function.cc:325-326."""
    a = U.HostArray(np.arange(4, dtype=np.int64), None, 0, 4).to_device(emu_ctx)
    g = U.HostArray(np.zeros(4, dtype=np.uint32), None, 0, 4).to_device(emu_ctx)
    with pytest.raises(emu_ctx.ArrowNotImplementedError, match="Direct execution of HASH_AGGREGATE"):
        emu_ctx.compute.call_function("hash_sum", [a, g])


@pytest.mark.parametrize("skip_nulls,min_count", [(True, 1), (False, 1), (True, 0), (True, 40), (False, 0)])
def test_hash_sum_kernel_vtable(emu_ctx, skip_nulls, min_count):
    """hash_sum(int64, uint32) through its HashAggregateKernel vtable: resize / consume (arrays,
    slices, broadcast + null scalars) / merge via group_id_mapping / finalize
    (hash_aggregate_numeric.cc:61-152; driven like groupby_aggregate_node.cc:210-337)."""
    P.check_hash_sum_kernel(emu_ctx, rng_for("hsk", skip_nulls, min_count), n=3000, num_groups=37,
                            skip_nulls=skip_nulls, min_count=min_count)


@pytest.mark.parametrize("num_groups,n", [(37, 6000), (5000, 20000), (300000, 20000)])
def test_hash_sum_kernel_partitioned_by_group_id(emu_ctx, num_groups, n):
    """The scratch form of the vtable consume (arx_hash_sum_i64_consume_ws) forced on: rows partitioned by the top bits
    of the dense group id, LDS aggregation, one flush per partition — no partition level (<= 2048 ids), one level,
    two levels; null values, hot groups, several consumes into the same state.  Same results as the per-row form."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
    try:
        P.check_hash_sum_kernel(emu_ctx, rng_for("hskp", num_groups), n=n, num_groups=num_groups, null_p=0.1,
                                use_pyarrow=False)
    finally:
        lib.arx_set_option(b"groupby_partition_min_rows", 1 << 17)


def test_hash_sum_kernel_no_nulls_has_no_bitmap(emu_ctx):
    P.check_hash_sum_kernel(emu_ctx, rng_for("hsk0"), n=1500, num_groups=11, null_p=0.0)


@pytest.mark.parametrize("mode", [0, 1])
def test_filter_forced_sweep_and_sparse_forms(emu_ctx, mode):
    """filter_sparse = 0 forces the sweeping compaction, 1 the gather form, for EVERY selectivity,
    null density, offset, width and length class: both must be bit-exact (auto picks by S/N)."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"filter_sparse", mode) == 0
    try:
        for n in (1, 63, 64, 65, 4095, 4096, 4097, 9000):
            for sel in ("drop", "emit_null"):
                rng = rng_for("fform", n, sel)
                v = U.random_array(rng, np.int64, n, null_p=0.1, offset=n % 5)
                m = U.random_mask(rng, n, 0.3, null_p=0.05, offset=n % 3)
                P.check_filter(emu_ctx, v, m, sel, use_pyarrow=False)
        for true_p in (0.0, 0.02, 0.5, 1.0):
            for vnull, mnull in ((0.0, 0.0), (0.2, 0.0), (0.0, 0.3), (1.0, 1.0)):
                for sel in ("drop", "emit_null"):
                    rng = rng_for("fform2", true_p, vnull, mnull, sel)
                    v = U.random_array(rng, np.int64, 20000, null_p=vnull, offset=7)
                    m = U.random_mask(rng, 20000, true_p, null_p=mnull, offset=13)
                    P.check_filter(emu_ctx, v, m, sel, use_pyarrow=False)
        for dtype in (np.int8, np.uint16, np.float32, np.float64):
            rng = rng_for("fform3", str(dtype))
            v = U.random_array(rng, dtype, 12345, null_p=0.1, offset=3)
            m = U.random_mask(rng, 12345, 0.15, null_p=0.05, offset=1)
            P.check_filter(emu_ctx, v, m, "emit_null", use_pyarrow=False)
    finally:
        lib.arx_set_option(b"filter_sparse", -1)


@pytest.mark.parametrize("l1_global,agg_chunk,bits", [(0, 1 << 16, 9), (1, 1 << 12, 9), (0, 1 << 20, 5), (1, 1 << 18, 11)])
def test_groupby_partition_knobs(emu_ctx, l1_global, agg_chunk, bits):
    """Tuning knobs of the partitioned consume never change results: level 1 with chunked exact offsets vs global
    cursors, the rows per LDS-aggregate work unit (how often a partition's groups are flushed)."""
    lib = emu_ctx._lib.get_lib()
    opts = {b"groupby_partition_min_rows": 0, b"groupby_partition_bits": bits, b"groupby_l1_global": l1_global,
            b"groupby_agg_chunk_rows": agg_chunk}
    for k_, v_ in opts.items():
        assert lib.arx_set_option(k_, v_) == 0
    try:
        rng = rng_for("gbpknobs", l1_global, agg_chunk, bits)
        n = 20000
        k = U.random_array(rng, np.int32, n, null_p=0.02, offset=3, lo=-2**31, hi=2**31 - 1)
        k.values[: n // 2] = k.values[: n // 2] % 1777
        v = U.random_array(rng, np.int64, n, null_p=0.1, offset=1)
        P.check_groupby_sum(emu_ctx, k, v, skip_nulls=True, min_count=1, batches=2, use_pyarrow=False)
        k2 = U.random_array(rng, np.int32, n, lo=0, hi=50000)
        v2 = U.random_array(rng, np.int64, n)
        P.check_groupby_sum(emu_ctx, k2, v2, use_pyarrow=False)
    finally:
        for k_, v_ in {b"groupby_partition_min_rows": 1 << 17, b"groupby_partition_bits": -1, b"groupby_l1_global": 1,
                       b"groupby_agg_chunk_rows": 1 << 18}.items():
            lib.arx_set_option(k_, v_)


@pytest.mark.parametrize("n,options", [(0, ()), (1, ()), (9000, ()), (30000, ((b"sort_msd", 1),)),
                                       (30000, ((b"sort_msd", 1), (b"sort_msd_sampled", 2))),
                                       (40000, ((b"sort_msd", 1), (b"sort_msd_segment_rows", 4096), (b"sort_msd_wide", 1), (b"sort_msd_wide_bits", 8))),
                                       (40000, ((b"sort_msd", 1), (b"sort_msd_segment_rows", 4096), (b"sort_msd_wide", 1), (b"sort_msd_wide_bits", 8),
                                                (b"sort_records_in_place", 0)))])
def test_sort_records(emu_ctx, n, options):
    """Round 6: arx_sort_records (the receiver of the sharded sort's records form) — LSD fallback, MSD hybrid, sampled
    splitters, the wide form on 12-byte records."""
    P.check_sort_records(emu_ctx, rng_for("sort-records", n, len(options)), n, options)


def test_groupby_range_state(emu_ctx):
    """Round 6: the range-partitioned state through its C ABI (plan / consume / merge / finalize)."""
    P.check_groupby_range_state(emu_ctx, rng_for)


def test_groupby_lines_plan(emu_ctx):
    """Round 6: the dense-range lines plan (write-combined whole-line scatter + direct-indexed LDS aggregate)."""
    P.check_groupby_lines_plan(emu_ctx, rng_for, wide_width=False)


@pytest.mark.parametrize("bits", [0, 1, 5, 9])   # (11 = a second two-level plan: GPU test only)
def test_groupby_partitioned_path(emu_ctx, bits):
    """The radix-partitioned consume (hist -> scatter level 1 [-> level 2] -> LDS aggregate -> flush)
    forced on, for one-level (bits <= 8) and two-level plans, with null keys / null values /
    wrap-around, several consume calls, and keys that overflow one partition's LDS table."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
    assert lib.arx_set_option(b"groupby_partition_bits", bits) == 0
    try:
        rng = rng_for("gbp", bits)
        n = 20000
        k = U.random_array(rng, np.int32, n, null_p=0.02, offset=3, lo=-2**31, hi=2**31 - 1)
        k.values[: n // 2] = k.values[: n // 2] % 1777          # many repeats + distinct tail
        v = U.random_array(rng, np.int64, n, null_p=0.1, offset=1)
        P.check_groupby_sum(emu_ctx, k, v, skip_nulls=(bits % 2 == 1), min_count=1, batches=2,
                            use_pyarrow=(bits == 9))
        # no nulls at all (the HAS_NULLS = false kernels)
        k2 = U.random_array(rng, np.int32, n, lo=0, hi=50000)
        v2 = U.random_array(rng, np.int64, n)
        P.check_groupby_sum(emu_ctx, k2, v2, use_pyarrow=False)
    finally:
        lib.arx_set_option(b"groupby_partition_min_rows", 1 << 17)
        lib.arx_set_option(b"groupby_partition_bits", -1)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("global_bits", [14, 4, -14])
def test_sort_msd_hybrid_path(emu_ctx, global_bits, fused):
    """The MSD-hybrid sort forced on (two global levels [+ the in-bucket level when the global
    bits are capped] + the windowed final ranking): full-range keys, heavy ties (bucket overflow ->
    LSD fallback), nulls (prep + MSD), descending, signed."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"sort_msd", 1) == 0
    assert lib.arx_set_option(b"sort_msd_fused", fused) == 0   # 1: LDS-resident bucket finish; 0: local + windowed final
    if global_bits < 0:   # the segmented form: an extra level on the top bits, then one pipeline per segment
        global_bits = -global_bits
        assert lib.arx_set_option(b"sort_msd_segment_rows", 4096) == 0
        # beyond the segment size the wide two-level form (run_msd_sort_wide) runs first; fused = 0 switches it off so
        # that the segmented form itself stays covered
        assert lib.arx_set_option(b"sort_msd_wide", fused) == 0
    assert lib.arx_set_option(b"sort_msd_global_bits", global_bits) == 0
    try:
        n = 14000   # (the emulator runs every workgroup as fibers on one core)
        rng = rng_for("msd", global_bits, fused)
        for dtype, order, placement, null_p in ((np.uint64, "ascending", "at_end", 0.0),
                                               (np.int64, "descending", "at_start", 0.03)):
            a = U.random_array(rng, dtype, n, null_p=null_p, offset=3)
            a.values[a.offset:a.offset + n - 1:5] = a.values[a.offset + 1:a.offset + n:5]  # ties
            P.check_sort_indices(emu_ctx, a, order, placement, use_pyarrow=(dtype == np.int64))
        ties = U.random_array(rng, np.uint64, n, lo=0, hi=7)       # 7 distinct keys: buckets overflow
        P.check_sort_indices(emu_ctx, ties, "ascending", "at_end", use_pyarrow=False)
        small = U.random_array(rng, np.uint64, 300)
        P.check_sort_indices(emu_ctx, small, "ascending", "at_end", use_pyarrow=False)
    finally:
        lib.arx_set_option(b"sort_msd", -1)
        lib.arx_set_option(b"sort_msd_global_bits", 14)
        lib.arx_set_option(b"sort_msd_segment_rows", 1 << 27)
        lib.arx_set_option(b"sort_msd_wide", 1)
        lib.arx_set_option(b"sort_msd_fused", 1)


@pytest.mark.parametrize("b2max", [12, 0])
@pytest.mark.parametrize("shift,gap2", [(2, 1), (2, 0), (0, 1), (0, 0)])
def test_sort_wide_sampled_level1(emu_ctx, shift, gap2, b2max):
    lib = emu_ctx._lib.get_lib()
    failed = P.check_sort_wide_sampled(emu_ctx, lib, rng_for("wide-sampled", shift, gap2), 300_000, shift, gap2, b2max)
    if shift == 0 and gap2 == 0:
        assert failed == 0   # exact counts never overflow
    if shift:
        # this sample misses the crafted inputs: both with the even split, at least one with the few large
        # level-1 buckets the 12-bit level 2 leaves at this size
        assert failed == 2 if b2max == 0 else failed >= 1


@pytest.mark.parametrize("rpt", [(8, 24), (16, 16)])   # (the default (24, 16) runs in the tests around this one)
def test_sort_wide_register_staged_tiles(emu_ctx, rpt):
    """Level-1 / level-2 scatter tiles of 8 (LDS-resident), 16 and 24 (register-staged) rows per thread: ragged last
    tiles, level-2 tiles that end inside a round of the LDS buffer, rooms that overflow (sorted / blocky inputs)."""
    lib = emu_ctx._lib.get_lib()
    # the 256-thread bucket finish takes over whenever every bucket fits it (always, at these sizes): one case without
    assert lib.arx_set_option(b"sort_msd_tiny_bucket", {16: 0, 8: 1}.get(rpt[0], 2)) == 0
    assert lib.arx_set_option(b"sort_msd_bucket_cpt", 8 if rpt[1] == 16 else 4) == 0   # sub-bucket counters per thread of the finish
    try:
        P.check_sort_wide_sampled(emu_ctx, lib, rng_for("wide-rpt", *rpt), 70_000, 2, 1, 12, rpt=rpt, typed_keys=True)
    finally:
        lib.arx_set_option(b"sort_msd_tiny_bucket", 2)
        lib.arx_set_option(b"sort_msd_bucket_cpt", 4)


@pytest.mark.parametrize("bits,b2max", [(13, 12)])
def test_sort_wide_many_level2_bins(emu_ctx, bits, b2max):
    # (each forced partition is a workgroup of fibers here: one shape, the default size mode — sampled level 1, fixed
    #  level-2 rooms; the GPU tier runs the rest)
    P.check_sort_wide_many_bins(emu_ctx, emu_ctx._lib.get_lib(), rng_for("wide-bins", bits, b2max), 30_000, bits,
                                b2max, combos=((2, 1),))


@pytest.mark.parametrize("n,bits,gap2,shift,rpt,b2max,wc,prefetch,l2w,wc_form", [
    (120_000, 0, 1, 0, (24, 16), 11, 256, 1, 1, 2), (40_000, 12, 0, 2, (8, 8), 11, 0, 1, 2, 2),
    (40_000, 6, 1, 2, (16, 24), 11, 1, 0, 0, 2), (120_000, 10, 1, 2, (24, 16), 4, 3, 1, 3, 2),
    (120_000, 10, 1, 2, (24, 16), 4, 2, 0, 1, 1), (160_000, 12, 0, 0, (8, 16), 3, 2, 1, 2, 2),
    (60_000, 12, 1, 2, (24, 16), 11, 4, 1, 3, 2)])    # (round 5's rank-and-stage level 1 at 160 000 rows: the GPU tier)
def test_sort_wide_rec8_words(emu_ctx, n, bits, gap2, shift, rpt, b2max, wc, prefetch, l2w, wc_form):
    """8-byte {key bits, row id} words through the wide form: ties below the word, duplicates, the tie budget, fall-backs;
    level 1 tile at a time and write-combined — round 6's append kernel (wc_form 2) and round 5's rank-and-stage kernel
    (2 to 512 bins, sampled and exact rooms, chunks of 4 / 2 / 1 lines, counted and fixed level-2 buckets)."""
    P.check_sort_wide_rec8(emu_ctx, emu_ctx._lib.get_lib(), rng_for("wide-rec8", bits, gap2), n, bits=bits, gap2=gap2, shift=shift,
                           rpt=rpt, b2max=b2max, wc=wc, prefetch=prefetch, l2w=l2w, wc_form=wc_form, wc_min_rows=1 << 13)


def test_null_count_bookkeeping(emu_ctx):
    P.check_null_count_bookkeeping(emu_ctx, rng_for("nullcount"))


@pytest.mark.parametrize("msd", [0, 1])
@pytest.mark.parametrize("dtype", [np.uint32, np.int32, np.float64, np.float32])
def test_sort_32bit_and_float_keys(emu_ctx, dtype, msd):
    """array_sort_indices on the other fixed-width key types: 32-bit integers (4 LSD passes), floats
    with NaNs as null-likes next to the nulls whatever the order, -0.0 tying with 0.0, infinities."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"sort_msd", 1 if msd else 0) == 0
    try:
        rng = rng_for("sort32f", str(dtype), msd)
        n = 4000   # (emulator: fibers on one core)
        for order, placement, null_p in (("ascending", "at_end", 0.05), ("descending", "at_start", 0.05)):
            a = U.random_array(rng, dtype, n, null_p=null_p, offset=2)
            v = a.values
            if np.dtype(dtype).kind == "f":
                v[::7] = np.nan
                v[::11] = 0.0
                v[1::11] = -0.0
                v[::13] = np.inf
                v[5::13] = -np.inf
                v[::3] = np.round(v[::3])          # ties
            else:
                v[::3] = v[::3] % 17               # ties
            P.check_sort_indices(emu_ctx, a, order, placement)
    finally:
        lib.arx_set_option(b"sort_msd", -1)


@pytest.mark.parametrize("dtype", [np.float64, np.uint64, np.int32])
def test_sort_msd_sampled_splitters(emu_ctx, dtype):
    """The sampled-splitter form of the MSD sort forced on: skewed keys (normal floats, clustered
    integers), duplicates-heavy columns (a bucket overflows -> LSD fallback), nulls, descending."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"sort_msd_sampled", 2) == 0
    try:
        rng = rng_for("sampled", str(dtype))
        n = 9000
        for order, placement, null_p in (("ascending", "at_end", 0.0), ("descending", "at_start", 0.04)):
            a = U.random_array(rng, dtype, n, null_p=null_p, offset=1)
            v = a.values
            if np.dtype(dtype).kind == "f":
                v[::50] = np.nan
                v[::17] = -0.0
            elif np.dtype(dtype) == np.uint64:
                v[:] = (np.abs(rng.standard_normal(len(v))) * 1e6).astype(np.uint64) + (v % 3) * 10**12   # clustered
            else:
                v[:] = (rng.standard_normal(len(v)) * 1000).astype(np.int32)                               # many ties
            P.check_sort_indices(emu_ctx, a, order, placement, use_pyarrow=(order == "ascending"))
        dup = U.random_array(rng, dtype, n, lo=None if np.dtype(dtype).kind == "f" else 0, hi=None if np.dtype(dtype).kind == "f" else 3)
        if np.dtype(dtype).kind == "f":
            dup.values[:] = np.round(dup.values)
        P.check_sort_indices(emu_ctx, dup, "ascending", "at_end", use_pyarrow=False)
    finally:
        lib.arx_set_option(b"sort_msd_sampled", 1)


# ------------------------------------------------------------------ binary / utf8 take + filter
@pytest.mark.parametrize("idx_dtype", [np.uint8, np.int16, np.uint32, np.int64])
@pytest.mark.parametrize("vnull,inull", [(0.0, 0.0), (0.2, 0.0), (0.0, 0.1), (0.3, 0.3)])
def test_binary_take(emu_ctx, idx_dtype, vnull, inull):
    """TestTakeKernelWithString (vector_selection_test.cc): values and index nulls, all index types."""
    rng = rng_for("btake", str(idx_dtype), vnull, inull)
    nv = 100 if np.dtype(idx_dtype).itemsize == 1 else 3000
    v = U.random_binary(rng, nv, null_p=vnull, offset=3, tail=2, utf8=True)
    i = U.random_array(rng, idx_dtype, 5000, null_p=inull, offset=1, lo=0, hi=nv - 1)
    P.check_binary_take(emu_ctx, v, i)


@pytest.mark.parametrize("m", [0, 1, 63, 64, 65, 4095, 4096, 4097, 8193])
def test_binary_take_lengths(emu_ctx, m):
    rng = rng_for("btakelen", m)
    v = U.random_binary(rng, 500, null_p=0.1, max_len=40)
    i = U.random_array(rng, np.int32, m, null_p=0.1, lo=0, hi=499)
    P.check_binary_take(emu_ctx, v, i)


def test_binary_take_empty_values_and_all_null(emu_ctx):
    rng = rng_for("btakeedge")
    v = U.random_binary(rng, 64, null_p=1.0)
    i = U.random_array(rng, np.int32, 200, lo=0, hi=63)
    out = P.check_binary_take(emu_ctx, v, i)
    assert out.null_count == 200
    v = U.random_binary(rng, 64, empty_p=1.0)      # only empty strings: zero data bytes
    P.check_binary_take(emu_ctx, v, i)


def test_binary_take_out_of_bounds(emu_ctx):
    rng = rng_for("btakeoob")
    v = U.random_binary(rng, 10)
    idx = U.HostArray(np.array([0, 3, 10, 2], dtype=np.int32), None, 0, 4)
    with pytest.raises(emu_ctx.ArrowIndexError, match="Index 10 out of bounds"):
        emu_ctx.compute.take(v.to_device(emu_ctx), idx.to_device(emu_ctx))


@pytest.mark.parametrize("sel", ["drop", "emit_null"])
@pytest.mark.parametrize("true_p,vnull,mnull", [(0.0, 0.1, 0.0), (0.3, 0.0, 0.0), (0.5, 0.2, 0.1), (1.0, 0.1, 0.05)])
def test_binary_filter(emu_ctx, sel, true_p, vnull, mnull):
    """TestFilterKernelWithString / FilterRandomTest on utf8 (vector_selection_test.cc)."""
    rng = rng_for("bfilter", sel, true_p, vnull, mnull)
    n = 6000
    v = U.random_binary(rng, n, null_p=vnull, offset=5, tail=3)
    m = U.random_mask(rng, n, true_p, null_p=mnull, offset=2, tail=1)
    P.check_binary_filter(emu_ctx, v, m, sel)


def test_binary_take_many_tiny_values(emu_ctx):
    """0/1-byte values: a 16 KiB output chunk spans many more rows than one LDS batch holds."""
    rng = rng_for("btaketiny")
    v = U.random_binary(rng, 4000, null_p=0.1, max_len=1, empty_p=0.5)
    i = U.random_array(rng, np.int32, 70000, null_p=0.05, lo=0, hi=3999)
    P.check_binary_take(emu_ctx, v, i)


def test_add_and_greater_with_a_scalar_operand(emu_ctx):
    P.check_scalar_operand_ops(emu_ctx, rng_for("scalarops"), n=5000)


@pytest.mark.parametrize("small_bucket,final_rows_log2,seg_min_bits,v2", [(0, 4, 1, 1), (1, 2, 1, 1), (1, 3, 5, 0), (0, 1, 1, 0)])
def test_sort_msd_bucket_variants_and_segment_fanout(emu_ctx, small_bucket, final_rows_log2, seg_min_bits, v2):
    """Tuning knobs of the MSD sort never change results: the 1024- vs 512-thread bucket kernel, the
    sub-bucket size and the fan-out of the segment level (segmented form forced on)."""
    lib = emu_ctx._lib.get_lib()
    opts = {b"sort_msd": 1, b"sort_msd_small_bucket": small_bucket, b"sort_msd_final_rows_log2": final_rows_log2,
            b"sort_msd_seg_min_bits": seg_min_bits, b"sort_msd_segment_rows": 2048 if seg_min_bits > 1 else 1 << 27,
            b"sort_msd_bucket_v2": v2, b"sort_msd_wide": v2}
    for k, v in opts.items():
        assert lib.arx_set_option(k, v) == 0
    try:
        rng = rng_for("msdknobs", small_bucket, final_rows_log2, seg_min_bits, v2)
        n = 6000   # (the emulator runs a 1024-thread workgroup as 1024 fibers)
        a = U.random_array(rng, np.uint64, n, null_p=0.02, offset=1)
        a.values[a.offset:a.offset + n - 1:7] = a.values[a.offset + 1:a.offset + n:7]  # ties
        P.check_sort_indices(emu_ctx, a, "descending", "at_start", use_pyarrow=False)
    finally:
        for k, v in {b"sort_msd": -1, b"sort_msd_small_bucket": 1, b"sort_msd_final_rows_log2": 1,
                     b"sort_msd_seg_min_bits": 1, b"sort_msd_segment_rows": 1 << 27, b"sort_msd_bucket_v2": 1, b"sort_msd_wide": 1}.items():
            lib.arx_set_option(k, v)


# ------------------------------------------------------------------ hash_min / hash_max on the fused table
@pytest.mark.parametrize("skip_nulls", [True, False])
@pytest.mark.parametrize("knull,vnull,batches", [(0.0, 0.0, 1), (0.05, 0.2, 3), (1.0, 0.5, 1), (0.1, 1.0, 2)])
def test_groupby_min_max(emu_ctx, skip_nulls, knull, vnull, batches):
    """GroupedMinMaxImpl (hash_aggregate.cc:330-419): extrema at the int64 edges, null keys as one
    group, groups whose values are all null, several consume calls."""
    rng = rng_for("gbminmax", skip_nulls, knull, vnull, batches)
    n = 4000
    k = U.random_array(rng, np.int32, n, null_p=knull, offset=2, lo=-40, hi=40)
    v = U.random_array(rng, np.int64, n, null_p=vnull, offset=1)
    v.values[5:9] = [2**63 - 1, -2**63, 0, -1]
    P.check_groupby_min_max(emu_ctx, k, v, skip_nulls, batches=batches)


@pytest.mark.parametrize("skip_nulls,min_count,knull,vnull,batches", [(True, 1, 0.0, 0.0, 1), (False, 1, 0.05, 0.2, 3),
                                                                        (True, 0, 0.1, 1.0, 2), (True, 3, 0.02, 0.5, 1)])
def test_groupby_mean_int64(emu_ctx, skip_nulls, min_count, knull, vnull, batches):
    """hash_mean(int64) (GroupedMeanImpl: doubles summed in row order) = (double)sum / count bit for bit while every
    partial sum is an exact integer; all-null groups, min_count = 0 (0 / 0 = NaN), !skip_nulls."""
    rng = rng_for("gmean", skip_nulls, min_count, knull, vnull, batches)
    n = 6000
    k = U.random_array(rng, np.int32, n, null_p=knull, offset=2, lo=-40, hi=40)
    v = U.random_array(rng, np.int64, n, null_p=vnull, offset=1, lo=-2**31, hi=2**31)
    P.check_groupby_mean(emu_ctx, k, v, skip_nulls, min_count, batches=batches)


def test_groupby_mean_declines_where_the_reference_is_order_dependent(emu_ctx):
    """Values near 2^62: the reference's double partial sums are rounded differently for different row orders,
    so there is nothing to be bit-exact with — NotImplemented with the reason, never a wrong number."""
    rng = rng_for("gmean-big")
    k = U.random_array(rng, np.int32, 5000, lo=0, hi=10)
    v = U.random_array(rng, np.int64, 5000, lo=2**60, hi=2**62)
    P.check_groupby_mean(emu_ctx, k, v, expect_decline=True)


@pytest.mark.parametrize("in_name", list(P.NUMERIC_TYPES))
def test_cast_every_numeric_pair(emu_ctx, in_name):
    """CastIntegerToInteger / CastFloatingToInteger / CastIntegerToFloating / CastFloatingToFloating
    (scalar_cast_numeric.cc:46-60, 190-207, 270-279) for all 10 x 10 numeric pairs, safe and unsafe."""
    rng = rng_for("castpair", in_name)
    for out_name in P.NUMERIC_TYPES:
        if out_name != in_name:
            P.check_cast_numeric_pair(emu_ctx, rng, in_name, out_name, n=3000)


@pytest.mark.parametrize("key_dtype", [np.int8, np.uint16, np.uint32, np.int64, np.uint64])
def test_groupby_sum_other_key_and_value_types(emu_ctx, key_dtype):
    rng = rng_for("gbtyped", str(key_dtype))
    for value_dtype in (np.int8, np.int32, np.uint32, np.uint64, np.int64):
        P.check_groupby_sum_typed(emu_ctx, rng, key_dtype, value_dtype, n=4000)


def test_groupby_declines_what_it_cannot_reproduce(emu_ctx):
    """64-bit keys beyond the int32 range (the device table holds 32-bit keys) and floating-point sums (row-order
    double accumulation in the reference) are NotImplemented with the reason — never a wrong answer."""
    amd = emu_ctx
    op = amd.compute.GroupBySum(64)
    with pytest.raises(NotImplementedError, match="beyond the int32 range"):
        op.consume(amd.Array.from_numpy(np.array([1, 2**40], dtype=np.int64)), amd.Array.from_numpy(np.array([1, 2], dtype=np.int64)))
    op = amd.compute.GroupBySum(64)
    with pytest.raises(NotImplementedError, match="row order"):
        op.consume(amd.Array.from_numpy(np.array([1, 2], dtype=np.int32)), amd.Array.from_numpy(np.array([1.0, 2.0])))


@pytest.mark.parametrize("dtype", [np.bool_, np.int8, np.uint16, np.int32, np.uint64, np.int64, np.float32, np.float64])
def test_indices_nonzero(emu_ctx, dtype):
    """IndicesNonZero (kernels/vector_selection.cc:352): uint64 positions of the valid, non-zero elements vs pyarrow."""
    import pyarrow as pa
    import pyarrow.compute as pc

    rng = rng_for("nonzero", str(dtype))
    for n in (0, 5, 70000, 100001):
        x = rng.integers(0, 3, n).astype(dtype) if dtype != np.bool_ else rng.random(n) < 0.3
        valid = rng.random(n) > 0.1
        got = emu_ctx.compute.indices_nonzero(emu_ctx.Array.from_numpy(x, valid if n else None))
        want = pc.indices_nonzero(pa.array(x, mask=(~valid) if n else None))
        assert got.type.name == "uint64" and np.array_equal(got.to_numpy()[0], want.to_numpy()), (dtype, n)


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.int32, np.uint32, np.float64, np.float32])
def test_golden_sort_and_sum_only_replay(emu_ctx, dtype):
    """The reference's own known-answer tests for the two sharded paths (vector_sort_test.cc:640-724,
    acero/hash_aggregate_test.cc:839-883; tests/golden/reference_vectors.json) replayed on the kernels."""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    assert P.replay_golden_sort(emu_ctx, gold, dtype) >= 20
    P.replay_golden_sum_only(emu_ctx, gold)


@pytest.mark.parametrize("run_end_type", ["int16", "int32", "int64"])
def test_filter_with_a_run_end_encoded_mask(emu_ctx, run_end_type):
    """array_filter(values, run_end_encoded<boolean>) (vector_selection_filter_internal.cc:1090): sliced masks, null run
    values, DROP and EMIT_NULL, vs the reference (pyarrow)."""
    import pyarrow as pa
    import pyarrow.compute as pc

    amd = emu_ctx
    rng = rng_for("reemask", run_end_type)
    ret = {"int16": pa.int16(), "int32": pa.int32(), "int64": pa.int64()}[run_end_type]
    for n in (1, 63, 64, 1000, 20000):
        if run_end_type == "int16" and n > 30000:
            continue
        lens = rng.integers(1, 200, max(8, n // 50))
        ends = np.cumsum(lens)
        ends = ends[ends < n + 300]
        if len(ends) == 0 or ends[-1] < n + 7:
            ends = np.append(ends, n + 7)
        rv = pa.array(rng.random(len(ends)) < 0.4, mask=rng.random(len(ends)) < 0.2)
        ree = pa.RunEndEncodedArray.from_arrays(pa.array(ends.astype(ret.to_pandas_dtype())), rv).slice(5, n)
        vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
        for sel in ("drop", "emit_null"):
            want = pc.filter(vals, ree, null_selection_behavior=sel)
            got = amd.compute.filter(amd.Array.from_pyarrow(vals), amd.array.RunEndEncoded.from_pyarrow(ree), sel)
            assert got.to_pyarrow().equals(want), (run_end_type, n, sel)


def test_groupby_min_max_next_to_sum(emu_ctx):
    rng = rng_for("gbminmaxsum")
    k = U.random_array(rng, np.int32, 3000, null_p=0.02, lo=0, hi=500)
    v = U.random_array(rng, np.int64, 3000, null_p=0.1, lo=-1000, hi=1000)
    P.check_groupby_min_max(emu_ctx, k, v, True, batches=2, with_sum=True)


def test_groupby_min_max_merge_of_two_states(emu_ctx):
    """Two states over halves of the rows, merged (Merge, hash_aggregate.cc:371-399) == one state."""
    amd = emu_ctx
    rng = rng_for("gbminmaxmerge")
    n = 3000
    k = U.random_array(rng, np.int32, n, null_p=0.03, lo=-60, hi=60)
    v = U.random_array(rng, np.int64, n, null_p=0.15)
    dk, dv = k.to_device(amd), v.to_device(amd)
    a, b = amd.compute.GroupBySum(1024, dk.device), amd.compute.GroupBySum(1024, dk.device)
    a.consume_min_max(dk.slice(0, 1700), dv.slice(0, 1700))
    b.consume_min_max(dk.slice(1700), dv.slice(1700))
    a.merge_min_max(b.export_min_max())
    gk, gkv, gmin, gmax, gvalid = (x.cpu().numpy() for x in a.finalize_min_max())
    w = O.groupby_minmax_i64(k.values, k.valid_bitmap(), 0, v.values, v.valid_bitmap(), 0, n, True)
    key = lambda r: (r[0] is None, r[0] or 0)  # noqa: E731
    got = sorted(((int(x) if y else None, (int(c), int(d)) if e else None) for x, y, c, d, e in zip(gk, gkv, gmin, gmax, gvalid)), key=key)
    want = sorted(((int(x) if y else None, (int(c), int(d)) if e else None)
                   for x, y, c, d, e in zip(w["keys"], w["key_is_valid"], w["mins"], w["maxs"], w["valid"])), key=key)
    assert got == want


@pytest.mark.parametrize("null_p,offset", [(0.0, 0), (0.05, 3)])
def test_unique_and_value_counts(emu_ctx, null_p, offset):
    """UniqueAction / ValueCountsAction (vector_hash.cc): first-appearance order from the fused table."""
    rng = rng_for("unique", null_p, offset)
    a = U.random_array(rng, np.int32, 3000, null_p=null_p, offset=offset, tail=2, lo=-200, hi=200)
    P.check_unique_and_value_counts(emu_ctx, a)          # (every sort launch costs seconds under the emulator)
    P.check_unique_and_value_counts(emu_ctx, U.random_array(rng, np.int32, 0))


@pytest.mark.parametrize("lnull,rnull,loff,roff", [(0.0, 0.0, 0, 0), (0.2, 0.0, 3, 0), (0.0, 0.3, 0, 65), (0.3, 0.3, 7, 13), (1.0, 0.5, 1, 2)])
def test_kleene_and_or_invert(emu_ctx, lnull, rnull, loff, roff):
    """KleeneAndOp / KleeneOrOp / InvertOp (scalar_boolean.cc): the full truth table incl. nulls,
    sliced operands with different bit offsets."""
    rng = rng_for("kleene", lnull, rnull, loff, roff)
    left = U.random_mask(rng, 5000, 0.5, null_p=lnull, offset=loff, tail=3)
    right = U.random_mask(rng, 5000, 0.4, null_p=rnull, offset=roff, tail=5)
    P.check_kleene_and_invert(emu_ctx, left, right)
    P.check_kleene_and_invert(emu_ctx, U.random_mask(rng, 0, 0.5), U.random_mask(rng, 0, 0.5))


def test_compare_family(emu_ctx):
    """Equal ... LessEqual (scalar_compare.cc:38-64): int64 and float64 incl. NaN / signed zeros / infinities."""
    P.check_compare_family(emu_ctx, rng_for("cmpfamily"), n=3000)


def test_subtract_multiply_and_checked_arithmetic(emu_ctx):
    """Subtract / Multiply / *Checked (base_arithmetic_internal.h): wrap-around vs "overflow" on valid slots only."""
    P.check_arithmetic(emu_ctx, rng_for("arith"), n=3000)


def test_integer_casts(emu_ctx):
    """CastIntegerToInteger (scalar_cast_numeric.cc:46-54) + IntegersInRange's first-offender message."""
    P.check_integer_casts(emu_ctx, rng_for("intcast"), n=4000)
    P.check_cast_i64_f64(emu_ctx, rng_for("i64f64"), n=4000)


@pytest.mark.parametrize("null_p,offset", [(0.0, 0), (0.07, 3)])
def test_dictionary_encode(emu_ctx, null_p, offset):
    """DictEncodeAction (vector_hash.cc:173-270): MASK and ENCODE null handling, first-appearance dictionary."""
    rng = rng_for("dictenc", null_p, offset)
    a = U.random_array(rng, np.int32, 2500, null_p=null_p, offset=offset, tail=2, lo=-150, hi=150)
    P.check_dictionary_encode(emu_ctx, a)
    P.check_dictionary_encode(emu_ctx, U.random_array(rng, np.int32, 0))


def test_scalar_aggregates_int64(emu_ctx):
    """SumImpl / CountImpl / MinMaxImpl (aggregate_basic.inc.cc): wrap-around sum, options, batches."""
    P.check_scalar_aggregates(emu_ctx, rng_for("scalaragg"), n=6000)


@pytest.mark.parametrize("idx_dtype", [np.uint8, np.int32, np.int64])
@pytest.mark.parametrize("vnull,inull,voff", [(0.0, 0.0, 0), (0.2, 0.1, 5), (1.0, 0.5, 67)])
def test_boolean_values_take_and_filter(emu_ctx, idx_dtype, vnull, inull, voff):
    """filter / take on BOOLEAN values (1-bit Gather, gather_internal.h; PrimitiveFilter's bit-width-1 case)."""
    rng = rng_for("booltake", str(idx_dtype), vnull, inull, voff)
    nv = 200 if np.dtype(idx_dtype).itemsize == 1 else 5000
    v = U.random_mask(rng, nv, 0.5, null_p=vnull, offset=voff, tail=3)
    i = U.random_array(rng, idx_dtype, 5000, null_p=inull, offset=1, lo=0, hi=nv - 1)
    m = U.random_mask(rng, nv, 0.3, null_p=0.05, offset=2)
    P.check_boolean_take_and_filter(emu_ctx, v, i, m)


@pytest.mark.parametrize("kind", ["int64", "int8", "bool", "utf8", "binary_nonull"])
def test_concat_arrays(emu_ctx, kind):
    """Concatenate (array/concatenate.cc): sliced chunks with and without validity, empty chunks, bit offsets
    that are not multiples of 8 / 64."""
    rng = rng_for("concat", kind)
    specs = [(77, 0.2, 3), (0, 0.0, 0), (1, 0.0, 0), (130, 0.0, 65), (64, 1.0, 7), (300, 0.1, 0), (5, 0.5, 1)]
    if kind == "bool":
        chunks = [U.random_mask(rng, n, 0.5, null_p=p, offset=o, tail=2) for n, p, o in specs]
    elif kind in ("utf8", "binary_nonull"):
        chunks = [U.random_binary(rng, n, null_p=0.0 if kind == "binary_nonull" else p, offset=o, tail=2, utf8=kind == "utf8")
                  for n, p, o in specs]
    else:
        chunks = [U.random_array(rng, np.dtype(kind).type, n, null_p=p, offset=o, tail=2) for n, p, o in specs]
    P.check_concat_arrays(emu_ctx, chunks)
    P.check_concat_arrays(emu_ctx, chunks[1:2])            # one empty chunk
    P.check_concat_arrays(emu_ctx, [chunks[3]])            # a single sliced chunk without nulls


@pytest.mark.parametrize("null_placement", ["at_end", "at_start"])
def test_order_by_several_keys(emu_ctx, null_placement):
    """OrderByNode::DoFinish with two and three sort keys of mixed direction (few distinct values per key so
    that every later key and the input order decide ties), a float key with NaNs, payload columns riding along."""
    rng = rng_for("orderby", null_placement)
    sizes = [(300, 3), (0, 0), (157, 0), (41, 5)]
    def col(make):
        return [make(n, o) for n, o in sizes]
    k0 = col(lambda n, o: U.random_array(rng, np.int32, n, null_p=0.1, offset=o, tail=1, lo=-3, hi=3))
    k1 = col(lambda n, o: U.random_array(rng, np.int64, n, null_p=0.1, offset=o, tail=1, lo=0, hi=4))
    def fkey(n, o):
        a = U.random_array(rng, np.float64, n, null_p=0.1, offset=o, tail=1)
        a.values[:] = np.round(a.values * 2) / 2
        a.values[rng.random(len(a.values)) < 0.1] = np.nan
        return a
    k2 = col(fkey)
    payload = col(lambda n, o: U.random_array(rng, np.int64, n, null_p=0.2, offset=o, tail=1))
    strs = col(lambda n, o: U.random_binary(rng, n, null_p=0.1, offset=o, tail=1, utf8=True))
    flags = col(lambda n, o: U.random_mask(rng, n, 0.5, null_p=0.1, offset=o, tail=1))
    cols = [k0, k1, k2, payload, strs, flags]
    # (every sort launch costs seconds under the emulator: the two placements share the key sets between them)
    other = "at_start" if null_placement == "at_end" else "at_end"
    if null_placement == "at_end":
        P.check_order_by(emu_ctx, cols, [(0, "ascending"), (1, "descending")], null_placement)
        P.check_order_by(emu_ctx, cols, [(1, "ascending")], null_placement)
    else:
        P.check_order_by(emu_ctx, cols, [(2, "descending"), (0, "descending"), (1, "ascending")], null_placement)
    P.check_order_by(emu_ctx, cols, [(0, "descending"), (2, "ascending")], [null_placement, other])     # per-key placement


def test_divide_and_divide_checked(emu_ctx):
    """Divide / DivideChecked (base_arithmetic_internal.h:366-424): truncating int64 division, zero divisors, INT64_MIN / -1,
    IEEE doubles, the last failing valid slot names the error, nulls hide failures."""
    P.check_divide(emu_ctx, rng_for("divide"), n=3000)


@pytest.mark.parametrize("dtype", ["int8", "uint8", "int16", "uint16", "int32", "uint32", "uint64", "float32"])
def test_divide_on_the_other_numeric_types(emu_ctx, dtype):
    """arx_divide_numeric: the same Call bodies per element type (min / -1 of the type's own width, unsigned types fail
    only on a zero divisor — DivideWithOverflowGeneric, util/int_util_overflow.h:124-138)."""
    P.check_divide(emu_ctx, rng_for("divide-" + dtype), n=1500, dtypes=(np.dtype(dtype),))


@pytest.mark.parametrize("section,combos", [("grouper_numeric_key", None), ("grouper_floating_point_key", None),
                                            ("grouper_multiple_int_keys", 40)])
def test_reference_grouper_golden_vectors(emu_ctx, section, combos):
    """grouper_test.cc's own expectations (exact ids in first-appearance order, uniques, Lookup nulls) on the device
    Grouper; the three-column case samples the type combinations (every width pair still occurs)."""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
    assert P.replay_golden_grouper(gold, section, P.device_grouper_factory(emu_ctx), combos) > 0


@pytest.mark.parametrize("dtypes,n,card,null_p,batches", [
    ((np.int64,), 5000, 700, 0.05, 1), ((np.int64,), 5000, 4999, 0.0, 3), ((np.uint64,), 3000, 5, 0.3, 2),
    ((np.int32, np.int32), 6000, 900, 0.1, 2), ((np.int64, np.int64), 4000, 300, 0.1, 1),
    ((np.int8, np.int16, np.int32, np.int64), 3000, 2500, 0.05, 4), ((np.float64, np.uint8), 2000, 50, 0.2, 1),
    ((np.int16,), 0, 1, 0.0, 1), ((np.uint8,) * 8, 2000, 100, 0.1, 2),
    # rows wider than one 16-byte table: the chain of tables (level s keys = id of level s-1 ++ the next columns)
    ((np.int64, np.int64, np.int64), 4000, 600, 0.1, 2), ((np.int64,) * 5, 3000, 2900, 0.05, 3),
    ((np.uint8,) * 20, 2500, 40, 0.2, 2), ((np.int32, np.int64, np.int16, np.float64, np.int8, np.int64), 3000, 30, 0.3, 1),
    ((np.int64, np.int64, np.int32), 0, 1, 0.0, 1)])
def test_grouper_ids_uniques_lookup(emu_ctx, dtypes, n, card, null_p, batches):
    P.check_grouper(emu_ctx, rng_for("grouper", len(dtypes), n, card, batches), dtypes, n, card, null_p, batches)


def test_grouper_hot_keys_take_the_pending_path(emu_ctx):
    """9 distinct rows over 30000: while a row's slot is claimed but not yet published (the emulator's __threadfence is
    a scheduling point) the other rows of the workgroup with that key must defer, not wait and not insert twice."""
    P.check_grouper(emu_ctx, rng_for("grouper-hot"), (np.int64, np.int64), 30000, 3, 0.0, 2, max_groups=16)


def test_grouper_declines_and_overflows(emu_ctx):
    from arrow_amd.array import int64, bool_

    with pytest.raises(NotImplementedError, match="1 to 32 key columns"):
        emu_ctx.compute.Grouper([int64] * 33, 16)
    with pytest.raises(NotImplementedError, match="keys of type bool"):
        emu_ctx.compute.Grouper([bool_], 16)
    g = emu_ctx.compute.Grouper([int64], 4)
    g.consume([emu_ctx.Array.from_numpy(np.arange(4, dtype=np.int64))])
    assert g.num_groups == 4
    with pytest.raises(ValueError, match="more than 4 distinct key rows"):
        g.consume([emu_ctx.Array.from_numpy(np.arange(10, dtype=np.int64))])


@pytest.mark.parametrize("n,null_p,offset,max_len,card", [(0, 0.0, 0, 8, 1), (3000, 0.1, 0, 30, 40), (2500, 0.0, 5, 11, 2000),
                                                         (2000, 0.3, 3, 50, 7), (1000, 1.0, 0, 5, 5), (1500, 0.1, 2, 700, 300)])
def test_binary_key_columns_and_first_rows(emu_ctx, n, null_p, offset, max_len, card):
    P.check_binary_key_columns(emu_ctx, U.random_binary_pool(rng_for("bkey", n, max_len), n, card, null_p, offset, max_len),
                               rng_for("bkey2", n))


def test_grouper_chain_levels_and_partial_lookups(emu_ctx):
    P.check_grouper_chain(emu_ctx)


@pytest.mark.parametrize("dtypes,n,card,null_p", [((np.int64,), 6000, 500, 0.05), ((np.int32, np.int32), 6000, 800, 0.1),
                                                  ((np.int64, np.int16), 4000, 3500, 0.0),
                                                  ((np.int64, np.int64, np.int64), 4000, 12, 0.1)])
def test_group_by_wide_and_multiple_keys(emu_ctx, dtypes, n, card, null_p):
    P.check_group_by_keys(emu_ctx, rng_for("group_by_keys", len(dtypes), n, card), dtypes, n, card, null_p)


@pytest.mark.parametrize("dtypes,n,m,idx_dtype", [
    ((np.int64, np.int32, np.float64), 5000, 3000, np.uint32),
    ((np.int8, np.int16, np.int64, np.uint64, np.float32), 777, 2000, np.int64),
    ((np.int64,) * 17, 300, 500, np.uint16),        # more than one group of 16 columns
    ((np.int32, np.int64), 1000, 0, np.int32), ((np.int64, np.int64), 64, 64, np.uint8)])
def test_take_record_batch_in_one_launch(emu_ctx, dtypes, n, m, idx_dtype):
    P.check_take_record_batch(emu_ctx, rng_for("take-rb", len(dtypes), n, m), dtypes, n, m, idx_dtype)


def test_take_record_batch_without_nulls_has_no_bitmaps(emu_ctx):
    rng = rng_for("take-rb-nonull")
    P.check_take_record_batch(emu_ctx, rng, (np.int64, np.int32), 4000, 4096, np.uint32, value_null_p=0.0, index_null_p=0.0,
                              offsets=False)


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64,
                                   np.float32, np.float64])
def test_compare_and_arithmetic_on_every_numeric_type(emu_ctx, dtype):
    P.check_numeric_compare_arith(emu_ctx, rng_for("numeric-ops", np.dtype(dtype).name), dtype, n=3000)


@pytest.mark.parametrize("wide", [0, 1])
def test_sort_keys_with_a_shared_prefix(emu_ctx, wide):
    lib = emu_ctx._lib.get_lib()
    P.check_sort_limited_range(emu_ctx, lib, rng_for("sort-prefix", wide), 9000 if wide else 5000, wide, light=not wide)


def test_compare_on_temporal_columns(emu_ctx):
    P.check_temporal_compare(emu_ctx, rng_for("temporal-compare"), n=3000)


def test_copy_segments_any_alignment(emu_ctx):
    P.check_copy_segments(emu_ctx, rng_for("copyseg"), 1)


@pytest.mark.parametrize("n,num_groups,null_p", [(4000, 37, 0.2), (2000, 5, 0.0), (300, 1, 1.0)])
def test_hash_minmax_and_count_dense_kernels(emu_ctx, n, num_groups, null_p):
    P.check_hash_minmax_count_kernels(emu_ctx, rng_for("hmmc", n, num_groups), n=n, num_groups=num_groups, null_p=null_p)


@pytest.mark.parametrize("dtype,n,num_groups,null_p", [(np.float64, 4000, 37, 0.2), (np.float32, 2000, 5, 0.0), (np.float64, 300, 1, 1.0)])
def test_hash_minmax_float_dense_kernels(emu_ctx, dtype, n, num_groups, null_p):
    P.check_hash_minmax_float_kernels(emu_ctx, rng_for("hmmf", n, num_groups), dtype=dtype, n=n, num_groups=num_groups, null_p=null_p)


def test_float_sum_is_the_references_bit_for_bit_and_float_min_max(emu_ctx):
    P.check_sum_float(emu_ctx, rng_for("fsum"), [0, 1, 15, 16, 17, 63, 64, 65, 1000, 4097, 16385, 40001])


def test_coalesce_of_two_operands_is_fill_null(emu_ctx):
    P.check_coalesce2(emu_ctx, rng_for("coalesce2"), n=3000)


def test_take_of_rows_of_any_width_and_of_lists_with_fixed_width_values(emu_ctx):
    """fixed_size_list / list / large_list selection where the nested values are fixed-width and free of nulls
    (FSLTakeExec -> FixedWidthTakeExec; ListSelectionImpl): arx_take_rows and arx_(large_)list_take_data."""
    P.check_take_rows(emu_ctx, rng_for("takerows"), n=700, m=600)
    P.check_list_take(emu_ctx, rng_for("listtake"), n=700, m=600)


def test_grouped_float_sum_is_the_references_row_order_sum(emu_ctx):
    """hash_sum / hash_mean of float32 / float64 values over dense group ids: the reference's row-order double accumulation
    per group, bit for bit (stable sort by group id + one walker per group)."""
    P.check_hash_sum_float(emu_ctx, rng_for("hashfsum"), n=3000, groups=(1, 7, 300))


def test_grouped_decimal128_sum(emu_ctx):
    """hash_sum / hash_min / hash_max of decimal128 values over dense group ids: 128-bit sums modulo 2^128 kept with two atomics
    per row; extrema in signed 128-bit order by one owner per group over the rows sorted by group id."""
    P.check_hash_sum_dec128(emu_ctx, rng_for("hashdec"), n=3000, groups=(1, 13, 400))
    P.check_hash_minmax_dec128(emu_ctx, rng_for("hashdecmm"), n=3000, groups=(1, 13, 400))
    P.check_reduce_dec128(emu_ctx, rng_for("reducedec"), sizes=(0, 1, 63, 64, 65, 5000))


def test_buffer_copy(emu_ctx):
    P.check_buffer_copy(emu_ctx, rng_for("bufcopy"), 1)


def test_bytes_to_bitmap(emu_ctx):
    P.check_bytes_to_bitmap(emu_ctx, rng_for("bytes-to-bitmap"))


def test_groupby_key_range(emu_ctx):
    P.check_groupby_key_range(emu_ctx, rng_for("key-range"))


def test_hash_any_all_dense_kernels(emu_ctx):
    P.check_hash_any_all_kernels(emu_ctx, rng_for("hash-bool"))


def test_bitmap_copy_segments(emu_ctx):
    P.check_bitmap_copy_segments(emu_ctx, rng_for("bitseg"), 1)


@pytest.mark.parametrize("bits", [1, 4, 8, 11])
def test_groupby_wide_one_level_form(emu_ctx, bits):
    """The wide one-level plan forced on (groupby_wide = 2): ONE flat scatter of register-held 24576-row tiles into
    2^bits bins of 12-byte records + 8192-slot LDS tables; null keys / null values / wrap-around, several consume
    calls, partial last tiles, more groups than the LDS tables hold (rows spill to the HBM table, still exact)."""
    lib = emu_ctx._lib.get_lib()
    opts = {b"groupby_partition_min_rows": 0, b"groupby_wide": 2, b"groupby_partition_bits": bits,
            b"groupby_wide_agg_chunk_rows": 1 << (14 + bits % 3)}
    for k_, v_ in opts.items():
        assert lib.arx_set_option(k_, v_) == 0
    wide0 = lib.arx_get_counter(b"groupby_slices_wide")
    try:
        rng = rng_for("gbwide", bits)
        n = 30000
        k = U.random_array(rng, np.int32, n, null_p=0.02, offset=3, lo=-2**31, hi=2**31 - 1)
        k.values[: n // 2] = k.values[: n // 2] % 1777          # many repeats + distinct tail
        v = U.random_array(rng, np.int64, n, null_p=0.1, offset=1)
        P.check_groupby_sum(emu_ctx, k, v, skip_nulls=(bits % 2 == 1), min_count=1, batches=2, use_pyarrow=(bits == 4))
        k2 = U.random_array(rng, np.int32, n + 4097, lo=0, hi=50000)      # no nulls: the HAS_NULLS = false kernels
        v2 = U.random_array(rng, np.int64, n + 4097)
        P.check_groupby_sum(emu_ctx, k2, v2, use_pyarrow=False)
        assert lib.arx_get_counter(b"groupby_slices_wide") >= wide0 + 3, "the wide plan did not run"
    finally:
        for k_, v_ in {b"groupby_partition_min_rows": 1 << 17, b"groupby_wide": 1, b"groupby_partition_bits": -1,
                       b"groupby_wide_agg_chunk_rows": 1 << 21}.items():
            lib.arx_set_option(k_, v_)


@pytest.mark.parametrize("wide,bits,parts", [(0, 0, 2), (0, 1, 3), (0, 5, 8), (0, 9, 4), (2, 4, 8), (2, 8, 5), (2, 6, 64)])
def test_groupby_consume_partials(emu_ctx, wide, bits, parts):
    """The sharded group-by's local pass without the local table (arx_groupby_sum_i64_consume_partials), on every plan of
    the partitioned consume: the unpartitioned aggregate, one- and two-level plans, the wide form with rooms and counted;
    several work units per partition (a key in several records), more groups than the LDS tables hold (rows that leave as
    records of their own), regions too small (ARX_CAPACITY_ERROR); shards with nulls or too small are declined."""
    lib = emu_ctx._lib.get_lib()
    opts = {b"groupby_partition_min_rows": 0, b"groupby_wide": wide or 1, b"groupby_partition_bits": bits,
            b"groupby_agg_chunk_rows": 1 << 12, b"groupby_wide_agg_chunk_rows": 1 << 14, b"groupby_wide_room_min_mean": 16}
    for k_, v_ in opts.items():
        assert lib.arx_set_option(k_, v_) == 0
    try:
        rng = rng_for("gbemit", wide, bits, parts)
        n = 40000
        k = U.random_array(rng, np.int32, n, lo=-2**31, hi=2**31 - 1)
        k.values[: n // 2] = k.values[: n // 2] % 1777          # many repeats + a distinct tail
        v = U.random_array(rng, np.int64, n, offset=1)
        records = P.check_groupby_consume_partials(emu_ctx, k, v, parts)
        assert records >= 1777 + n // 2 - 64
        k2 = U.random_array(rng, np.int32, n + 4097, lo=0, hi=50000, offset=5)
        v2 = U.random_array(rng, np.int64, n + 4097)
        P.check_groupby_consume_partials(emu_ctx, k2, v2, parts)
        # declined: nulls, and a shard below the partitioned consume's row threshold
        kn = U.random_array(rng, np.int32, n, null_p=0.01, lo=0, hi=100)
        from arrow_amd import parallel
        assert parallel.consume_partials_regions(kn.to_device(emu_ctx), v.to_device(emu_ctx), 1 << 12, parts) is None
        assert lib.arx_set_option(b"groupby_partition_min_rows", 1 << 17) == 0
        assert parallel.consume_partials(k.to_device(emu_ctx), v.to_device(emu_ctx), 1 << 12, parts) == (None, None)
        # more records than the regions hold (every row its own group and a capacity that promises few groups: the rows that
        # find no place in an LDS table leave as records of their own): declined, the caller goes through the table
        assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
        big = 400_000 if bits == 0 else 0
        if big:
            kd = U.random_array(rng, np.int32, big, lo=-2**31, hi=2**31 - 1)
            vd = U.random_array(rng, np.int64, big)
            assert parallel.consume_partials(kd.to_device(emu_ctx), vd.to_device(emu_ctx), 1 << 12, parts) == (None, None)
            P.check_groupby_sum(emu_ctx, kd, vd, capacity=1 << 20, use_pyarrow=False)       # (the table path still serves it)
        k16 = U.random_array(rng, np.int16, n, lo=0, hi=100)      # other key types: through the table (its casts)
        assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
        assert parallel.consume_partials_regions(k16.to_device(emu_ctx), v.to_device(emu_ctx), 1 << 12, parts) is None
    finally:
        for k_, v_ in {b"groupby_partition_min_rows": 1 << 17, b"groupby_wide": 1, b"groupby_partition_bits": -1,
                       b"groupby_agg_chunk_rows": 1 << 18, b"groupby_wide_agg_chunk_rows": 1 << 21,
                       b"groupby_wide_room_min_mean": 1 << 14}.items():
            lib.arx_set_option(k_, v_)


GROUPBY_STRIPE_DEFAULT = 0   # arrow_amd/csrc/groupby.hip g_gbp_stripe


@pytest.mark.parametrize("stripe", [0, 4, 68])
@pytest.mark.parametrize("hot", [False, True])
def test_groupby_wide_form_without_histogram(emu_ctx, hot, stripe):
    """The wide plan with fixed rooms instead of a histogram pass: evenly spread keys fit their rooms (no overflow);
    a few hot keys outgrow one room — the slice is redone with counted partitions (null rows are not consumed twice)
    and the call's later slices stay on the counted plan.  Results exact either way — and whether the rooms lie one after
    the other or in stripes of `stripe` records (groupby_stripe)."""
    lib = emu_ctx._lib.get_lib()
    opts = {b"groupby_partition_min_rows": 0, b"groupby_wide": 2, b"groupby_partition_bits": 6, b"groupby_wide_room_min_mean": 16, b"groupby_stripe": stripe,
            b"groupby_wide_max_slice_rows": 98304}
    for k_, v_ in opts.items():
        assert lib.arx_set_option(k_, v_) == 0
    names = (b"groupby_slices_rooms", b"groupby_rooms_overflows")
    before = [lib.arx_get_counter(c) for c in names]
    try:
        rng = rng_for("gbrooms", hot)
        n = 250000
        k = U.random_array(rng, np.int32, n, null_p=0.02, lo=-2**31, hi=2**31 - 1)
        if hot:
            k.values[n // 3:] = 7            # two thirds of the rows in ONE group: its partition outgrows its room
        v = U.random_array(rng, np.int64, n, null_p=0.1)
        P.check_groupby_sum(emu_ctx, k, v, skip_nulls=False, min_count=2, batches=1, use_pyarrow=not hot)
    finally:
        for k_, v_ in {b"groupby_partition_min_rows": 1 << 17, b"groupby_wide": 1, b"groupby_partition_bits": -1,
                       b"groupby_wide_room_min_mean": 1 << 14, b"groupby_wide_max_slice_rows": (1 << 32) - (1 << 26),
                       b"groupby_stripe": GROUPBY_STRIPE_DEFAULT}.items():
            lib.arx_set_option(k_, v_)
    rooms, overflows = (lib.arx_get_counter(c) - b for c, b in zip(names, before))
    assert rooms >= 1, "the plan without a histogram did not run"
    assert (overflows >= 1) == hot, (rooms, overflows)
    if hot:
        assert rooms == 1, "after an overflow the call stays on the counted plan"


@pytest.mark.parametrize("distinct", [900, 0])
def test_groupby_probe_slice_selects_the_plan(emu_ctx, distinct):
    """A capacity that only bounds the group count from above (two-level plan) + enough rows: a HyperLogLog sketch of
    the first groupby_probe_rows keys estimates the distinct keys; few of them, seen often -> the rows run the wide plan, keys
    that do not repeat -> the two-level plan.  Same groups either way."""
    lib = emu_ctx._lib.get_lib()
    assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
    assert lib.arx_set_option(b"groupby_probe_rows", 4096) == 0
    assert lib.arx_set_option(b"groupby_wide_max_bits", 3) == 0    # (so that the capacity bound alone cannot pick the wide plan)
    names = (b"groupby_slices_probe", b"groupby_slices_wide", b"groupby_slices_two_level")
    before = [lib.arx_get_counter(c) for c in names]
    try:
        rng = rng_for("gbprobe", distinct)
        n = 9 * (4096) + 1234
        hi = distinct if distinct else 2**31 - 1
        k = U.random_array(rng, np.int32, n, null_p=0.01, lo=-5 if distinct else -2**31, hi=hi)
        v = U.random_array(rng, np.int64, n, null_p=0.05)
        P.check_groupby_sum(emu_ctx, k, v, capacity=1 << 21, batches=1, use_pyarrow=False)
        P.check_groupby_sum(emu_ctx, k, v, capacity=1 << 21, batches=2, use_pyarrow=False)   # second consume: table not empty
    finally:
        lib.arx_set_option(b"groupby_partition_min_rows", 1 << 17)
        lib.arx_set_option(b"groupby_probe_rows", 1 << 25)
        lib.arx_set_option(b"groupby_wide_max_bits", 11)
    probe, wide, two = (lib.arx_get_counter(c) - b for c, b in zip(names, before))
    assert probe == 3, "one sketch (round 3: one probe slice) per consume call"
    if distinct:
        assert (wide, two) == (3, 0), (probe, wide, two)     # round 4: the sketch aggregates nothing, every row runs the wide plan
    else:
        assert (wide, two) == (0, 3), (probe, wide, two)


def test_hash_product_group_edge_rows_and_dec128_split(emu_ctx):
    """The C-ABI entry points behind hash_product / hash_first / hash_last / hash_one and the decimal128 sort keys against the
    oracle's restatements of GroupedProductImpl, GroupedFirstLastImpl and GroupedOneImpl."""
    P.check_hash_product_and_edge_rows(emu_ctx, rng_for("hashprod"), n=3000, groups=(1, 7, 300))



def test_group_moments_variance_stddev_skew_kurtosis(emu_ctx):
    """hash_variance / hash_stddev / hash_skew / hash_kurtosis: the two-pass moments kernels against the oracle's restatement
    of GroupedStatisticImpl (kernels/hash_aggregate_numeric.cc:457-843)."""
    P.check_group_moments(emu_ctx, rng_for("moments"), n=3000, groups=(1, 7, 300))

def test_rank(emu_ctx):
    """Round 6 (f3): rank / rank_quantile = arx_sort_indices + arx_rank against the oracle's restatement of vector_rank.cc and the
    known answers of the reference's TestRank."""
    P.check_rank(emu_ctx, rng_for, light=True)


def test_select_k_and_partition_nth(emu_ctx):
    """Round 6 (f3): select_k_unstable / partition_nth_indices on the sort skeleton — the promised properties."""
    P.check_select_k_partition_nth(emu_ctx, rng_for, light=True)


def test_sort_boolean_keys(emu_ctx):
    """Round 6 (f3): boolean sort keys — the counting sort as three GetTakeIndices."""
    P.check_sort_boolean_keys(emu_ctx, rng_for)
