"""CPU-only check of the Arrow registration shim (arrow_amd/csrc/arrow_plugin.cc): with
ARROW_AMD_PLUGIN_DRY_RUN=1 the kernels are added to Arrow's live registry without a device.
Inputs below the staging threshold are handed to Arrow's own stock kernels (results unchanged);
a call that is routed to the HIP path must fail loudly — there is no CPU compute in the shim."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    from arrow_amd.plugin_build import build_plugin
    so = build_plugin()
    a = pa.array(np.arange(1000), mask=np.arange(1000) % 7 == 0)
    m = pa.array(np.arange(1000) % 3 == 0)
    f = pa.array(np.linspace(-1e40, 1e40, 1000))
    g = pa.array(np.linspace(1e40, -1e40, 1000), mask=np.arange(1000) % 5 == 0)
    fn = pa.array(f.to_numpy(), mask=np.arange(1000) % 11 == 0)
    def run():
        # greater(double, double) is re-registered with NO_PREALLOCATE flags: every shape the
        # ScalarExecutor used to prepare for the stock kernel must come out identical
        gt = [pc.greater(fn, g), pc.greater(fn.slice(3, 500), g.slice(7, 500)), pc.greater(f, g), pc.greater(fn, f),
              pc.greater(fn, 0.5), pc.greater(0.5, g), pc.greater(fn, pa.scalar(None, pa.float64())),
              pc.greater(pa.scalar(1.0), pa.scalar(0.5)), pc.greater(pa.scalar(1.0), pa.scalar(None, pa.float64())),
              pc.greater(pa.chunked_array([fn.slice(0, 300), fn.slice(300)]), g),
              pc.greater(pa.array([], pa.float64()), pa.array([], pa.float64())),
              pa.table({"x": fn, "y": g}).filter(pc.field("x") > pc.field("y")).column("x").combine_chunks()]
        a2 = pa.array(np.arange(1000)[::-1].copy(), mask=np.arange(1000) % 4 == 0)
        big = pa.array(np.array([2**63 - 1, -2**63, 5], dtype=np.int64))
        gt += [pc.add(a, a2), pc.add(a.slice(3, 400), a2.slice(9, 400)), pc.add(a, 5), pc.add(5, a2), pc.add(big, big),
               pc.add(a, pa.scalar(None, pa.int64())), pc.add(fn, g), pc.add(fn, 0.5), pc.add(f, g),
               pc.add(pa.chunked_array([a.slice(0, 300), a.slice(300)]), a2), pc.add(pa.scalar(1), pa.scalar(2)),
               pc.add(pa.array(np.arange(10, dtype=np.int32)), pa.array(np.arange(10, dtype=np.int64))),   # implicit cast
               pc.greater(a, a2), pc.greater(a, 500), pc.greater(500, a2), pc.greater(a.slice(1, 10), a2.slice(2, 10)),
               pc.greater(pa.array([], pa.int64()), pa.array([], pa.int64()))]
        import decimal
        ts = pa.array(np.arange(1000) * 1000, pa.timestamp("us", tz="UTC"), mask=np.arange(1000) % 6 == 0)
        dec = pa.array([None if i % 8 == 0 else decimal.Decimal(i * 1001) / 1000 for i in range(1000)], pa.decimal128(20, 3))
        fsb16 = pa.array([bytes([i % 251] * 16) for i in range(1000)], pa.binary(16))
        fsb3 = pa.array([bytes([i % 251] * 3) for i in range(1000)], pa.binary(3))
        wide = [ts, dec, fsb16, fsb3, pa.array(np.arange(1000, dtype=np.int32), pa.time32("s")),
                pa.array(np.arange(1000), pa.duration("ns")), pa.array(np.arange(1000, dtype=np.float16))]
        ix = pa.array([5, 1, 999, None, 0], pa.int16())
        for sv in (ts, pa.array(np.arange(1000, dtype=np.int32)[::-1].copy(), pa.date32()),
                   pa.array(((np.arange(1000) * 7919) % 86400).astype(np.int32), pa.time32("s")), pa.array(-np.arange(1000), pa.duration("ms")),
                   pa.array(np.arange(1000)[::-1].copy(), pa.date64()), pa.array(np.arange(1000) % 17, pa.time64("us"))):
            gt += [pc.array_sort_indices(sv), pc.array_sort_indices(sv, order="descending", null_placement="at_start")]
        for wv in wide:
            gt += [pc.filter(wv, m), pc.filter(wv, m, null_selection_behavior="emit_null"), pc.take(wv, ix)]
        small = pa.array((np.arange(1000) % 1000) - 500, mask=np.arange(1000) % 6 == 0)
        for ar in (pc.subtract, pc.multiply, pc.add_checked, pc.subtract_checked, pc.multiply_checked):
            gt += [ar(small, a2), ar(small.slice(2, 300), a2.slice(5, 300)), ar(small, 3), ar(3, small), ar(fn, g), ar(fn, 0.5),
                   ar(small, pa.scalar(None, pa.int64())), ar(pa.chunked_array([small.slice(0, 400), small.slice(400)]), a2)]
        gt += [pc.subtract(big, pa.array(np.array([-1, 1, 0], dtype=np.int64))), pc.multiply(big, big),
               pc.add_checked(pa.array([2**63 - 1, 5], mask=np.array([True, False])), pa.array([1, 1])),   # the overflowing slot is null
               pa.table({"x": a, "y": a2}).filter((pc.field("x") - pc.field("y") * 2) < 0).column("x").combine_chunks()]
        d2 = pc.add(a2, 1)      # (no zero divisors)
        gt += [pc.divide(a, d2), pc.divide_checked(a.slice(2, 300), d2.slice(5, 300)), pc.divide(a, 7), pc.divide(1000, d2), pc.divide(fn, g),
               pc.divide(pa.array([-2**63, 7, -7]), pa.array([-1, 2, 2]))]
        for bad in (lambda: pc.add_checked(big, big), lambda: pc.multiply_checked(big, 2), lambda: pc.subtract_checked(-2, big),
                    lambda: pc.divide(a, 0), lambda: pc.divide_checked(pa.array([-2**63, 1]), pa.array([-1, 1])),
                    lambda: pc.divide_checked(pa.array([1.5]), pa.array([0.0]))):
            try:
                bad()
            except pa.lib.ArrowInvalid as e:
                gt.append(str(e))
            else:
                raise SystemExit("expected overflow")
        for cmp in (pc.equal, pc.not_equal, pc.greater_equal, pc.less, pc.less_equal):
            gt += [cmp(a, a2), cmp(a.slice(2, 300), a2.slice(5, 300)), cmp(a, 500), cmp(500, a2), cmp(fn, g), cmp(fn, 0.0),
                   cmp(0.25, g), cmp(a, pa.scalar(None, pa.int64())), cmp(pa.chunked_array([fn.slice(0, 400), fn.slice(400)]), g),
                   cmp(pa.array([float("nan"), 1.0, -0.0]), pa.array([float("nan"), 1.0, 0.0]))]
        bm = pa.array(np.arange(1000) % 2 == 0, mask=np.arange(1000) % 9 == 0)
        gt += [pc.and_kleene(m, bm), pc.or_kleene(m, bm), pc.and_kleene(bm.slice(3, 500), m.slice(70, 500)),
               pc.and_kleene(bm, True), pc.or_kleene(False, bm), pc.and_kleene(bm, pa.scalar(None, pa.bool_())),
               pc.or_kleene(pa.scalar(True), pa.scalar(None, pa.bool_())), pc.invert(bm), pc.invert(m.slice(5)),
               pc.invert(pa.scalar(None, pa.bool_())), pc.and_kleene(pa.chunked_array([bm.slice(0, 300), bm.slice(300)]), m),
               pa.table({"x": fn, "y": g}).filter((pc.field("x") > pc.field("y")) & ~(pc.field("x") > 0.5)).column("x").combine_chunks()]
        strs = strs_h = pa.array([None if i % 5 == 0 else "s" * (i % 7) for i in range(1000)])
        gt += [pc.filter(strs, m), pc.take(strs, pa.array([5, 1, 999, None])),
               pc.filter(strs.cast(pa.binary()), m, null_selection_behavior="emit_null")]
        # scalar aggregates of HOST int64 data pass through our shim to the stock state (incl. chunked input, options, Acero)
        an = pa.array(np.arange(1000) * 7 - 3000, mask=np.arange(1000) % 11 == 0)
        for opts in (None, pc.ScalarAggregateOptions(skip_nulls=False), pc.ScalarAggregateOptions(min_count=995)):
            gt += [pc.call_function(fname, [arr], opts) for fname in ("sum", "min_max", "min", "max") for arr in
                   (an, a, an.slice(5, 0), pa.chunked_array([an.slice(0, 300), an.slice(300)]))]
        gt += [pc.count(an, mode=mode) for mode in ("only_valid", "only_null", "all")] + [pc.count(strs_h), pc.sum(f), pc.mean(an)]
        gt += [pa.table({"x": an}).group_by([]).aggregate([("x", "sum"), ("x", "min_max"), ("x", "count"), ("x", "max")]).to_pydict().__repr__()]
        return gt + [pc.filter(a, m), pc.take(a, pa.array([5, 1, 999])), pc.greater(f, pa.array(f.to_numpy()[::-1].copy())),
                pc.array_sort_indices(pa.array(np.arange(1000)[::-1].astype(np.uint64))),
                pc.array_sort_indices(pa.array((np.arange(1000) % 13).astype(np.int64), mask=np.arange(1000) % 9 == 0),
                                      order="descending", null_placement="at_start"),
                pc.sort_indices(pa.table({"a": pa.array((np.arange(1000) % 7).astype(np.uint64))}), sort_keys=[("a", "descending")]),
                pc.array_sort_indices(pa.chunked_array([pa.array(np.arange(10, dtype=np.uint64)), pa.array(np.arange(5, dtype=np.uint64))])),
                pc.array_sort_indices(pa.array([1.5, float("nan"), None, -0.0, 0.0, float("inf"), float("nan")]), order="descending"),
                pc.array_sort_indices(pa.array(np.arange(300, dtype=np.int32)[::-1].copy())),
                pc.array_sort_indices(pa.array(np.linspace(-3, 3, 200).astype(np.float32)), null_placement="at_start"),
                pc.cast(f, pa.float32(), safe=False), pc.cast(f.slice(3), pa.int64(), safe=False)]
    before = pc.get_function("array_filter").num_kernels
    stock = run()
    os.environ["ARROW_AMD_PLUGIN_DRY_RUN"] = "1"
    lib = ctypes.CDLL(so)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    assert lib.arrow_amd_register() == 0            # idempotent
    assert pc.get_function("array_filter").num_kernels > before
    ours = run()
    for i, (x, y) in enumerate(zip(stock, ours)):
        assert (x == y) if isinstance(x, str) else x.equals(y), (i, x, y)
        if isinstance(x, pa.Array):
            assert x.null_count == y.null_count, i
    for fn in (b"array_filter", b"array_take", b"greater", b"array_sort_indices", b"cast", b"add", b"boolean", b"compare", b"reduce"):
        assert lib.arrow_amd_plugin_calls(fn, 0) >= 1, fn     # handed to Arrow's stock kernel
        assert lib.arrow_amd_plugin_calls(fn, 1) == 0, fn     # nothing claimed to be a GPU call
    assert lib.arrow_amd_plugin_calls(b"no_such_function", 0) == -1
    import torch
    if not torch.cuda.is_available():
        # routed to the HIP path without a device: must raise, not compute somewhere else
        lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(10))
        lib.arrow_amd_plugin_set_min_rows_streaming(ctypes.c_int64(10))
        for call in (lambda: pc.filter(a, m), lambda: pc.cast(f, pa.float32(), safe=False),
                     lambda: pa.table({"k": pa.array([1, 2, 1], pa.int32()), "v": pa.array([1, 2, 3], pa.int64())})
                     .group_by("k").aggregate([("v", "sum")])):
            try:
                call()
            except (OSError, pa.ArrowException) as e:
                assert "HIP" in str(e) or "hip" in str(e), str(e)
            else:
                raise SystemExit("a HIP-path call succeeded without a GPU")
    # the Acero factory of the fused group-by is registered under a NEW name (duplicates are
    # rejected, acero/exec_plan.cc:1132-1142); without a device running it must fail loudly
    from pyarrow import acero
    tab = pa.table({"k": pa.array([1, 2, 1], pa.int32()), "v": pa.array([1, 2, 3], pa.int64())})
    decl = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))])
    if not torch.cuda.is_available():
        try:
            decl.to_table()
        except (OSError, pa.ArrowException) as e:
            assert "HIP" in str(e) or "hip" in str(e), str(e)
        else:
            raise SystemExit("aggregate_rocm ran without a GPU")
        try:
            acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
                acero.Declaration("order_by_rocm", acero.OrderByNodeOptions([("v", "descending")]))]).to_table()
        except (OSError, pa.ArrowException) as e:
            assert "HIP" in str(e) or "hip" in str(e), str(e)
        else:
            raise SystemExit("order_by_rocm ran without a GPU")
    print("REGISTRATION_OK")
''')


def test_registration_dry_run():
    pytest.importorskip("pyarrow")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + SCRIPT], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "REGISTRATION_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
