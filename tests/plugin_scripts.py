"""The scripts of the plugin tier and the ONE table both tiers consume.

Each script registers libarrow_amd_plugin.so on Arrow's own live FunctionRegistry in a fresh interpreter and checks
unmodified pyarrow.compute / Acero calls against the stock build's answers.  `CASES` is the only list of plugin tests:
tests/test_gpu_arrow_plugin.py runs every row on the MI355X, tests/test_plugin_emulated.py runs every row under the CPU
emulator at `emu_scale` — no per-test Python body exists outside the scripts, so the CPU gate executes every line the
GPU tier will (VERDICT r4 "Next round" 1b)."""
import os
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    so = build_plugin()
    rng = np.random.default_rng(5)
    n = SC(1_000_003)
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    mask = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.05)
    idx = pa.array(rng.integers(0, n, SC(300_000)).astype(np.int32), mask=rng.random(SC(300_000)) < 0.1)
    f64a, f64b = pa.array(rng.standard_normal(n)), pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1)
    keys = pa.array(rng.integers(0, 2**63, n).astype(np.uint64), mask=rng.random(n) < 0.05)
    small = pa.array(np.arange(100))
    gtab = pa.table({"k": pa.array(rng.integers(-500, 500, n).astype(np.int32), mask=rng.random(n) < 0.01),
                     "v": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.1)})
    def run():
        return dict(
            f_drop=pc.filter(vals, mask), f_emit=pc.filter(vals, mask, null_selection_behavior="emit_null"),
            f_slice=pc.filter(vals.slice(3), mask.slice(3)),
            f_i32=pc.filter(vals.cast(pa.int64()).slice(0, SC(200_000)).cast(pa.int32(), safe=False), mask.slice(0, SC(200_000))),
            take=pc.take(vals, idx), take_nb=pc.take(vals, idx, boundscheck=False),
            table=pa.table({"v": vals, "w": f64a}).filter(mask),
            gt=pc.greater(f64a, f64b), sort=pc.array_sort_indices(keys),
            sort_d=pc.array_sort_indices(keys, order="descending", null_placement="at_start"),
            sort_f64=pc.array_sort_indices(f64b, order="descending", null_placement="at_start"),
            sort_i32=pc.array_sort_indices(vals.slice(0, SC(400_000)).cast(pa.int64()).cast(pa.int32(), safe=False)),
            small=pc.filter(small, pa.array(np.arange(100) % 2 == 0)),
            boolv=pc.filter(mask, mask),
            cast=pc.cast(f64b, pa.float32(), safe=False), cast_slice=pc.cast(f64b.slice(5), pa.float32()),
            cast_other=pc.cast(f64a.slice(0, 5000), pa.int64(), safe=False),
            gb=gtab.group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k"),
            gb_threads=gtab.group_by("k", use_threads=True).aggregate(
                [("v", "sum", pc.ScalarAggregateOptions(skip_nulls=False, min_count=2))]).sort_by("k"),
        )
    stock = run()
    lib = ctypes.CDLL(so)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_gpu_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_stock_calls.restype = ctypes.c_int64
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(1000))
    # element-wise kernels on host arrays stay on the CPU by default (PCIe moves more than the CPU
    # computes); force the staging path so that it is covered too
    lib.arrow_amd_plugin_set_min_rows_streaming(ctypes.c_int64(1000))
    ours = run()
    gpu_calls = lib.arrow_amd_plugin_gpu_calls()
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    fns = ["array_filter", "array_take", "greater", "array_sort_indices", "cast", "hash_sum"]
    stats = {f: (lib.arrow_amd_plugin_calls(f.encode(), 1), lib.arrow_amd_plugin_calls(f.encode(), 0)) for f in fns}
    for k in stock:
        assert ours[k].equals(stock[k]), (k, stats)
        if hasattr(ours[k], "null_count"):
            assert ours[k].null_count == stock[k].null_count, k
    # every large call above took the HIP path: (gpu, stock) calls per function
    want_gpu = {"array_filter": 4, "array_take": 4, "greater": 1, "array_sort_indices": 4, "cast": 2,
                "hash_sum": 2}
    for f, wmin in want_gpu.items():
        assert stats[f][0] >= wmin, (f, stats)
    assert stats["array_filter"][1] >= 1, stats   # the tiny input was handed to the stock kernel (so are host-resident boolean values)
    assert stats["cast"][1] >= 1, stats           # float64 -> int64 is not ours: stock meta-function
    try:
        pc.take(vals, pa.array(np.array([0, n, 1] * 1000, dtype=np.int64)))
        raise SystemExit("expected IndexError")
    except pa.lib.ArrowIndexError as e:
        assert str(e) == f"Index {n} out of bounds", str(e)
    print("PLUGIN_OK gpu_calls=%d stock_calls=%d %r" % (gpu_calls, lib.arrow_amd_plugin_stock_calls(), stats))
''')


HASH_KERNELS_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        pa.set_cpu_count(1)      # the SIMT emulator runs one kernel at a time: thread-local states and Merge still happen
        pa.set_io_thread_count(1)  # (use_threads=True), their consumes just never overlap
    rng = np.random.default_rng(31)
    n = SC(1_500_000)
    k = pa.array(rng.integers(-400, 400, n), mask=rng.random(n) < 0.01)              # int64 keys: the stock CPU Grouper
    v = pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.15)
    big = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.15)        # partial sums beyond 2^53
    s = pa.array([("x%d" % (i % 97)) if i % 7 else None for i in range(n)])
    f = pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.3)
    # booleans for hash_any / hash_all: mostly-true and mostly-false columns (so that both outcomes occur per group), 15 % nulls
    bt = pa.array(rng.random(n) < 0.995, mask=rng.random(n) < 0.15)
    bf = pa.array(rng.random(n) < 0.005, mask=rng.random(n) < 0.15)
    # hash_sum over every integer width: narrow values are widened on the device, unsigned sums come back as uint64
    # (full-range uint64 / int64 sums wrap modulo 2^64 exactly as the reference's do)
    ints = {"i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=rng.random(n) < 0.1),
            "u16": pa.array(rng.integers(0, 2**16, n).astype(np.uint16)),
            "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=rng.random(n) < 0.1),
            "u32": pa.array(rng.integers(0, 2**32, n).astype(np.uint32), mask=rng.random(n) < 0.5),
            "u64": pa.array(rng.integers(0, 2**64, n, dtype=np.uint64), mask=rng.random(n) < 0.1)}
    t = pa.table({"k": k, "v": v, "big": big, "s": s, "f": f, "bt": bt, "bf": bf, **ints})
    tc = pa.concat_tables([t.slice(0, n // 3), t.slice(n // 3, n // 5), t.slice(n // 3 + n // 5)])   # several chunks
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    aggs = [("v", "min"), ("v", "max"), ("v", "mean"), ("big", "mean"), ("v", "count"),
            ("v", "count", pc.CountOptions(mode="only_null")), ("v", "count", pc.CountOptions(mode="all")),
            ("s", "count"), ("f", "count", pc.CountOptions(mode="only_null")),
            ("v", "min", strict), ("v", "max", strict), ("v", "mean", strict), ("v", "sum"),
            ("bt", "any"), ("bt", "all"), ("bf", "any"), ("bf", "all"),
            ("bt", "any", strict), ("bt", "all", strict), ("bf", "any", strict), ("bf", "all", strict),
            ("bf", "all", pc.ScalarAggregateOptions(skip_nulls=True, min_count=1600)),
            ("i8", "sum"), ("u16", "sum"), ("i32", "sum"), ("u32", "sum"), ("u64", "sum"), ("i32", "sum", strict), ("u32", "sum", strict),
            ("i8", "min"), ("i8", "max"), ("u16", "min"), ("u16", "max"), ("i32", "min"), ("i32", "max", strict), ("u32", "min", strict), ("u32", "max"),
            ("i8", "mean"), ("u16", "mean"), ("i32", "mean"), ("u32", "mean", strict)]
    def run(tab, threads):
        return tab.group_by("k", use_threads=threads).aggregate(aggs).sort_by("k")
    # ---- the reference kernels first: registering the plugin re-routes these very calls
    want = {(name, threads): run(tab, threads) for name, tab in (("t", t), ("tc", tc)) for threads in (False, True)}
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    gpu0, stock0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1), lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    for (name, threads), w in want.items():
        got = run(t if name == "t" else tc, threads)
        assert got.schema.equals(w.schema), (got.schema, w.schema)
        for col in range(w.num_columns):      # (by position: several aggregates of one column share their name)
            if threads and w.schema.names[col] == "big_mean":
                # the reference's own threaded answer moves in the last bits from run to run here: doubles accumulated
                # per thread in row order, merged in completion order, sums beyond 2^53
                a, b = (np.asarray(x.column(col).combine_chunks().fill_null(0)) for x in (got, w))
                assert got.column(col).is_null().equals(w.column(col).is_null()) and np.allclose(a, b, rtol=1e-12, atol=0), (name, "big_mean")
                continue
            assert got.column(col).equals(w.column(col)), (name, threads, w.schema.names[col], got.column(col).slice(0, 5), w.column(col).slice(0, 5))
    gpu1, stock1 = lib.arrow_amd_plugin_calls(b"hash_sum", 1), lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    assert gpu1 - gpu0 >= 4 * 10, ("the hash_* vtables did not run on the device", gpu0, gpu1)
    assert stock1 > stock0, "hash_mean of HOST values is the reference kernel's (bit-exact whatever the magnitudes)"

    # ---- device-resident VALUE columns under the stock GroupByNode (host keys -> CPU Grouper -> ids; values in HBM)
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    m = SC(400_000)
    th = t.slice(0, m).combine_chunks()
    td = pa.table({"k": th.column("k").chunk(0), "v": to_device(th.column("v").chunk(0)),
                   "i32": to_device(th.column("i32").chunk(0)), "u64": to_device(th.column("u64").chunk(0)),
                   "bt": to_device(th.column("bt").chunk(0))})
    daggs = [("v", "hash_min", None, "mn"), ("v", "hash_max", None, "mx"), ("v", "hash_mean", None, "me"),
             ("v", "hash_count", None, "c"), ("v", "hash_count", pc.CountOptions(mode="only_null"), "cn"),
             ("v", "hash_sum", None, "sm"), ("i32", "hash_sum", None, "s32"), ("u64", "hash_sum", None, "s64"),
             ("i32", "hash_min", None, "mn32"), ("i32", "hash_max", None, "mx32"), ("i32", "hash_mean", None, "me32"),
             ("u64", "hash_min", None, "mn64"), ("u64", "hash_max", strict, "mx64"),     # (full-range uint64: on x + 2^63 as int64)
             ("bt", "hash_any", None, "any"), ("bt", "hash_all", strict, "all")]
    def plan(tab):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("aggregate", acero.AggregateNodeOptions(daggs, keys=["k"]))]).to_table(use_threads=False).sort_by("k")
    wd = want[("t", False)]     # not the same rows: recompute the expectation on the slice with the (now plugged) host route,
    wh = plan(pa.table({"k": th.column("k"), "v": th.column("v"), "i32": th.column("i32"), "u64": th.column("u64"),
                        "bt": th.column("bt")}))   # which was just shown equal to the reference
    stock2 = lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    gd = plan(td)
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock2, "device-resident values must not reach a reference kernel"
    for col in wh.schema.names:
        assert gd.column(col).equals(wh.column(col)), (col, gd.column(col).slice(0, 5), wh.column(col).slice(0, 5))
    # partial sums beyond 2^53 on the device route: refused loudly, not approximated
    tb = pa.table({"k": th.column("k").chunk(0), "v": to_device(th.column("big").chunk(0))})
    try:
        acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tb)),
            acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "hash_mean", None, "me")], keys=["k"]))]).to_table(use_threads=False)
        raise SystemExit("hash_mean beyond 2^53 on device values did not fail")
    except pa.ArrowNotImplementedError as e:
        assert "2^53" in str(e), e
    print("HASH_KERNELS_OK")
''')


VECTOR_HASH_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(41)
    n = SC(800_000)
    def col(dtype, card, null_p, typ=None):
        if np.issubdtype(dtype, np.floating):
            pool = rng.standard_normal(card).astype(dtype)
            pool[: min(4, card)] = [0.0, -0.0, np.nan, np.inf][: min(4, card)]      # bits compared: 0.0 and -0.0 stay apart
        else:
            info = np.iinfo(dtype)
            pool = rng.integers(info.min, info.max, card, dtype=dtype, endpoint=True)
        v = pool[rng.integers(0, card, n)]
        return pa.array(v, type=typ, mask=(rng.random(n) < null_p) if null_p else None)
    cases = {"i64": col(np.int64, 5000, 0.05), "i32": col(np.int32, 70000, 0.0), "u8": col(np.uint8, 200, 0.2),
             "i16": col(np.int16, 3, 0.5), "u64": col(np.uint64, n, 0.01), "f64": col(np.float64, 900, 0.1),
             "f32": col(np.float32, 50, 0.0), "ts": col(np.int64, 1000, 0.1, pa.timestamp("us")),
             "d32": col(np.int32, 1000, 0.02, pa.date32()), "allnull": pa.array([None] * 100, pa.int32()),
             "empty": pa.array([], pa.int64()), "nullfirst": pa.array([None, 5, None, 7, 5, 9], pa.int64())}
    enc = pc.DictionaryEncodeOptions(null_encoding="encode")
    # ---- the reference kernels first (registering re-routes the device calls only, but keep the order honest)
    want = {}
    for name, a in cases.items():
        sl = a.slice(3) if len(a) > 10 else a
        want[name] = dict(u=pc.unique(a), vc=pc.value_counts(a), de=pc.dictionary_encode(a), dee=pc.dictionary_encode(a, options=enc),
                          us=pc.unique(sl), des=pc.dictionary_encode(sl), dn=pc.drop_null(a), dns=pc.drop_null(sl))
        if not pa.types.is_temporal(a.type):
            want[name].update(nz=pc.indices_nonzero(a), nzs=pc.indices_nonzero(sl))
    ca = pa.chunked_array([cases["i64"].slice(0, n // 3), cases["i64"].slice(n // 3, n // 2), cases["i64"].slice(n // 3 + n // 2)])
    want_chunked = dict(u=pc.unique(ca), vc=pc.value_counts(ca), de=pc.dictionary_encode(ca))
    boolean = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.1)
    want_bool = dict(nz=pc.indices_nonzero(boolean), nzs=pc.indices_nonzero(boolean.slice(5)), dn=None)
    num_types = [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64(), pa.float32(), pa.float64()]
    small = pa.array(rng.integers(0, 100, n), mask=rng.random(n) < 0.1)        # fits every numeric type
    want_cast = {(str(a), str(b)): pc.cast(pc.cast(small, a), b) for a in num_types for b in num_types}
    wide = pa.array(rng.integers(-2**40, 2**40, n))
    fr = pa.array(rng.standard_normal(n) * 1000)
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(m) for m in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    def same(h, w):      # Array.equals with NaN == NaN (floats compare by their bits, like the kernels)
        if h.type != w.type or len(h) != len(w):
            return False
        if pa.types.is_floating(h.type):
            bits = pa.int64() if h.type == pa.float64() else pa.int32()
            return h.view(bits).equals(w.view(bits))
        if pa.types.is_struct(h.type):
            return all(same(h.field(i), w.field(i)) for i in range(h.type.num_fields))
        if pa.types.is_dictionary(h.type):
            return h.indices.equals(w.indices) and same(h.dictionary, w.dictionary)
        return h.equals(w)

    def check(tag, got, w):
        assert not got.is_cpu, ("output left the device", tag)
        h = to_host(got)
        h.validate(full=True)
        assert same(h, w), (tag, h.slice(0, 8), w.slice(0, 8), len(h), len(w))

    gpu0 = lib.arrow_amd_plugin_calls(b"unique", 1)
    for name, a in cases.items():
        d = to_device(a)
        ds = d.slice(3) if len(a) > 10 else d
        w = want[name]
        check(name + " unique", pc.unique(d), w["u"])
        check(name + " unique (sliced)", pc.unique(ds), w["us"])
        check(name + " value_counts", pc.value_counts(d), w["vc"])
        check(name + " dictionary_encode", pc.dictionary_encode(d), w["de"])
        check(name + " dictionary_encode (sliced)", pc.dictionary_encode(ds), w["des"])
        check(name + " dictionary_encode(encode)", pc.dictionary_encode(d, options=enc), w["dee"])
        if a.null_count or len(a) == 0:
            # (by name: drop_null is a MetaFunction, replaced in the registry; pyarrow's generated pc.drop_null wrapper
            #  keeps the Function object it found at import time and would run the reference on HBM addresses)
            check(name + " drop_null", pc.call_function("drop_null", [d]), w["dn"])
            check(name + " drop_null (sliced)", pc.call_function("drop_null", [ds]), w["dns"])
        if not pa.types.is_temporal(a.type):
            check(name + " indices_nonzero", pc.indices_nonzero(d), w["nz"])
            check(name + " indices_nonzero (sliced)", pc.indices_nonzero(ds), w["nzs"])
    assert lib.arrow_amd_plugin_calls(b"unique", 1) - gpu0 >= 6 * (len(cases) - 2), "the hash vector kernels did not run on the device"
    stock0 = lib.arrow_amd_plugin_calls(b"unique", 0)
    assert pc.unique(cases["i64"]).equals(want["i64"]["u"]) and lib.arrow_amd_plugin_calls(b"unique", 0) == stock0 + 1   # host: the reference kernel
    # several device chunks: one Grouper across the chunks (it grows by re-consuming its own uniques)
    dca = pa.chunked_array([to_device(c) for c in ca.chunks])
    check("chunked unique", pc.unique(dca), want_chunked["u"])
    check("chunked value_counts", pc.value_counts(dca), want_chunked["vc"])
    got = pc.dictionary_encode(dca)
    assert got.num_chunks == want_chunked["de"].num_chunks
    for g_chunk, w_chunk in zip(got.chunks, want_chunked["de"].chunks):
        check("chunked dictionary_encode", g_chunk, w_chunk)
    d_bool = to_device(boolean)
    check("bool indices_nonzero", pc.indices_nonzero(d_bool), want_bool["nz"])
    check("bool indices_nonzero (sliced)", pc.indices_nonzero(d_bool.slice(5)), want_bool["nzs"])
    # ---- every numeric cast pair on device arrays (arx_cast_numeric), and the reference's refusals
    c0 = lib.arrow_amd_plugin_calls(b"cast", 1)
    d_small = {str(a): to_device(pc.cast(small, a)) for a in num_types}
    for a in num_types:
        for b in num_types:
            if a == b:
                continue
            check(f"cast {a}->{b}", pc.cast(d_small[str(a)], b), want_cast[(str(a), str(b))])
            check(f"cast {a}->{b} (sliced)", pc.cast(d_small[str(a)].slice(9), b), want_cast[(str(a), str(b))].slice(9))
    assert lib.arrow_amd_plugin_calls(b"cast", 1) - c0 >= 150
    d_wide, d_fr = to_device(wide), to_device(fr)
    for src, host, target in ((d_wide, wide, pa.int16()), (d_fr, fr, pa.int32()), (d_wide, wide, pa.float32())):
        try:
            pc.cast(host, target)
            raise SystemExit("the reference accepted this cast?")
        except pa.ArrowInvalid as e:
            ref_msg = str(e)
        try:
            pc.cast(src, target)
            raise SystemExit(f"unsafe device cast to {target} did not fail")
        except pa.ArrowInvalid as e:
            assert str(e).split(" ")[0:2] == ref_msg.split(" ")[0:2], (str(e), ref_msg)
        check(f"unsafe cast to {target}", pc.cast(src, target, safe=False), pc.cast(host, target, safe=False))
    print("VECTOR_HASH_OK")
''')


GENERAL_GROUP_BY_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(51)
    n = SC(2_000_000)
    t = pa.table({
        "k64": pa.array(rng.integers(-2**62, 2**62, 5000)[rng.integers(0, 5000, n)], mask=rng.random(n) < 0.01),   # the whole int64 range
        "a": pa.array(rng.integers(-30, 30, n).astype(np.int32), mask=rng.random(n) < 0.02),
        "b": pa.array(rng.integers(0, 40, n).astype(np.int16)),
        "c": pa.array(rng.integers(0, 3, n).astype(np.uint8), mask=rng.random(n) < 0.1),
        "d": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32()),
        # (hash_mean on the device is exact while rows x max|v| of a group stays below 2^53 — the null-key group of k64
        #  holds 1 % of the rows)
        "v": pa.array(rng.integers(-2**36, 2**36, n), mask=rng.random(n) < 0.15),
        "w": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.05),
        "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2),
        # narrower integer VALUE columns (widened on the device; unsigned sums are uint64, extrema keep the column's type)
        "i16": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16), mask=rng.random(n) < 0.1),
        "u32": pa.array(rng.integers(0, 2**32, n).astype(np.uint32), mask=rng.random(n) < 0.3),
        "u64": pa.array(rng.integers(0, 2**64, n, dtype=np.uint64), mask=rng.random(n) < 0.1),
        "u64s": pa.array(rng.integers(0, 2**36, n, dtype=np.uint64), mask=rng.random(n) < 0.1),      # (means stay exact)
        # booleans for hash_any / hash_all: rarely true with nulls, almost always true without (so both outcomes occur per group)
        "rare": pa.array(rng.random(n) < 0.01, mask=rng.random(n) < 0.1),
        "sure": pa.array(rng.random(n) < 0.97),
    })
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    plans = [
        (["d"], [("rare", "hash_any", None, "any"), ("rare", "hash_all", strict, "all"), ("sure", "hash_all", None, "sall"),
                 ("sure", "hash_any", pc.ScalarAggregateOptions(skip_nulls=True, min_count=90), "sany"), ("flag", "hash_all", strict, "fall"),
                 ("v", "hash_sum", None, "s")]),
        (["k64"], [("v", "hash_sum", None, "s"), ("w", "hash_sum", None, "sw"), ("v", "hash_min", None, "mn"), ("v", "hash_max", None, "mx"),
                   ("v", "hash_mean", None, "me"), ("v", "hash_count", None, "c"), ("flag", "hash_count", pc.CountOptions(mode="only_null"), "fn"),
                   ([], "hash_count_all", None, "all")]),
        (["a", "b"], [("v", "hash_sum", strict, "s"), ("v", "hash_mean", strict, "me"), ("w", "hash_max", strict, "mx"), ("d", "hash_count", pc.CountOptions(mode="all"), "c")]),
        (["d", "b", "c"], [("w", "hash_sum", None, "s"), ("v", "hash_min", None, "mn")]),
        (["a"], [("v", "hash_sum", None, "s"), ("w", "hash_sum", None, "sw")]),      # int32 key but two value columns: not the fused operator's case
        (["a", "c"], [("i16", "hash_sum", None, "s16"), ("i16", "hash_min", None, "mn16"), ("i16", "hash_max", strict, "mx16"), ("i16", "hash_mean", None, "me16"),
                      ("u32", "hash_sum", strict, "s32"), ("u32", "hash_min", None, "mn32"), ("u32", "hash_max", None, "mx32"), ("u32", "hash_mean", strict, "me32"),
                      ("u64", "hash_sum", None, "s64"), ("u64", "hash_min", None, "mn64"), ("u64", "hash_max", strict, "mx64"),
                      ("u64s", "hash_mean", None, "me64"), ("u64s", "hash_min", strict, "mn64s")]),
    ]
    def run(tab, node, keys, aggs):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=False).sort_by([(k, "ascending") for k in keys])
    # ---- the reference GroupByNode with the reference kernels, before anything is registered
    want = [run(t, "aggregate", keys, aggs) for keys, aggs in plans]
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def same(got, w, tag):
        assert got.schema.equals(w.schema), (tag, got.schema, w.schema)
        for i in range(w.num_columns):
            assert got.column(i).equals(w.column(i)), (tag, w.schema.names[i], got.column(i).slice(0, 5), w.column(i).slice(0, 5))

    # several chunks so that several batches arrive; device-resident: every column in HBM, sliced chunks (offsets != 0)
    chunks = [t.slice(0, n // 3), t.slice(n // 3, 7), t.slice(n // 3 + 7)]
    tc = pa.concat_tables(chunks)
    # (unsliced arrays: pyarrow's ChunkedArray constructor counts the nulls of a sliced array on the CPU; table_source
    #  slices the chunks into <= 32Ki-row batches itself, so offsets != 0 arrive anyway)
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 2 + 3), t.slice(n // 2 + 3))])
    td_host = t
    g0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    for (keys, aggs), w in zip(plans, want):
        same(run(tc, "aggregate_rocm", keys, aggs), w, ("host", keys))
        same(run(td, "aggregate_rocm", keys, aggs), run(td_host, "aggregate_rocm", keys, aggs), ("device vs host", keys))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - g0 >= 3 * len(plans), "aggregate_rocm did not run the device Grouper"
    # value types it does not take: refused by name (key rows wider than one 16-byte Grouper table go through a chain of
    # tables: tests/test_gpu_group_keys.py)
    for keys, aggs, needle in ((["a", "b"], [("flag", "hash_sum", None, "s")], "integer or floating-point values"),):
        try:
            run(t, "aggregate_rocm", keys, aggs)
            raise SystemExit("aggregate_rocm accepted " + str(keys))
        except pa.ArrowNotImplementedError as e:
            assert needle in str(e), e
    # an empty input still produces the schema
    e = run(t.slice(0, 0), "aggregate_rocm", ["a", "b"], [("v", "hash_sum", None, "s")])
    assert e.num_rows == 0 and e.schema.names == ["a", "b", "s"], e.schema
    print("GENERAL_GROUP_BY_OK")
''')


DEVICE_INTERFACES_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    emulated = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1"
    if emulated:      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_sync_event_waits.restype = ctypes.c_int64
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(m) for m in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    rng = np.random.default_rng(61)
    n = 200_000 if emulated else 5_000_000
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    mask = pa.array(rng.random(n) < 0.2)
    d_vals, d_mask = to_device(vals), to_device(mask)
    # ---- Device::Stream + SyncEvent: a producer that does NOT synchronise hands its array over with sync_event set;
    # the importer's buffers carry the event and the shims make their stream wait for it before the kernels read
    c_dev, c_schema, o_dev, o_schema = (ctypes.create_string_buffer(m) for m in (128, 72, 128, 72))
    d_vals._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_on_stream_with_event(c_dev, c_schema, o_dev, o_schema) == 0, lib.arrow_amd_plugin_last_error()
    sync_event = ctypes.cast(ctypes.addressof(o_dev) + 96, ctypes.POINTER(ctypes.c_void_p))[0]     # ArrowDeviceArray.sync_event
    assert sync_event, "the exported array carries no sync event"
    evented = pa.Array._import_from_c_device(ctypes.addressof(o_dev), ctypes.addressof(o_schema))
    w0 = lib.arrow_amd_plugin_sync_event_waits()
    got = pc.filter(evented, d_mask)
    assert lib.arrow_amd_plugin_sync_event_waits() > w0, "the imported buffers' sync event was not waited for"
    assert to_host(got).equals(pc.filter(vals, mask))
    # ---- MemoryManager::GetBufferWriter / GetBufferReader
    src = rng.integers(0, 256, 1_000_003, dtype=np.uint8)
    dst = np.zeros_like(src)
    assert lib.arrow_amd_device_buffer_round_trip(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(src)),
                                                  dst.ctypes.data_as(ctypes.c_void_p)) == 0, lib.arrow_amd_plugin_last_error()
    assert (src == dst).all()
    # ---- DLPack: refusals everywhere; the zero-copy hand-over to PyTorch-ROCm on the GPU tier
    lib.arrow_amd_export_dlpack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    def dlpack(darr):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        out = ctypes.c_void_p()
        rc = lib.arrow_amd_export_dlpack(ctypes.addressof(c_dev), ctypes.addressof(c_schema), ctypes.byref(out))
        return rc, out, lib.arrow_amd_plugin_last_error()
    rc, _, msg = dlpack(d_vals)
    assert rc != 0 and b"no nulls" in msg, msg
    rc, _, msg = dlpack(d_mask)
    assert rc != 0 and b"DLPack" in msg, msg
    dense = pa.array(rng.integers(-2**62, 2**62, n))
    d_dense = to_device(dense)
    c_probe, c_probe_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    d_dense._export_to_c_device(ctypes.addressof(c_probe), ctypes.addressof(c_probe_schema))
    base = ctypes.c_void_p.from_address(ctypes.c_void_p.from_address(ctypes.addressof(c_probe) + 40).value + 8).value   # ArrowArray.buffers[1]
    rc, ptr, msg = dlpack(d_dense.slice(5))
    assert rc == 0, msg
    class DLTensor(ctypes.Structure):
        _fields_ = [("data", ctypes.c_void_p), ("device_type", ctypes.c_int32), ("device_id", ctypes.c_int32), ("ndim", ctypes.c_int32),
                    ("code", ctypes.c_uint8), ("bits", ctypes.c_uint8), ("lanes", ctypes.c_uint16), ("shape", ctypes.POINTER(ctypes.c_int64)),
                    ("strides", ctypes.c_void_p), ("byte_offset", ctypes.c_uint64)]
    t = DLTensor.from_address(ptr.value)
    assert (t.device_type, t.ndim, t.code, t.bits, t.lanes, t.shape[0], t.byte_offset, t.data) == (10, 1, 0, 64, 1, n - 5, 0, base + 40), \
        (t.device_type, t.ndim, t.code, t.bits, t.lanes, t.shape[0], t.byte_offset, t.data, base)   # the slice is in the pointer
    if emulated:
        # nobody takes the capsule here: call the deleter ourselves (DLManagedTensor.deleter follows manager_ctx)
        deleter = ctypes.cast(ctypes.c_void_p.from_address(ptr.value + ctypes.sizeof(DLTensor) + 8).value, ctypes.CFUNCTYPE(None, ctypes.c_void_p))
        deleter(ptr.value)
    else:
        import torch
        ctypes.pythonapi.PyCapsule_New.restype = ctypes.py_object
        ctypes.pythonapi.PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
        capsule = ctypes.pythonapi.PyCapsule_New(ptr, b"dltensor", None)
        tensor = torch.from_dlpack(capsule)
        assert tensor.is_cuda and tensor.dtype == torch.int64 and tensor.numel() == n - 5
        assert tensor.data_ptr() == base + 40, "not zero-copy"
        assert torch.equal(tensor.cpu(), torch.from_numpy(dense.to_numpy()[5:]))
        # and the tensor outlives the Arrow array: the buffers stay alive until the tensor is dropped
        del d_dense
        assert int(tensor[123].item()) == int(dense[128].as_py())
    print("DEVICE_INTERFACES_OK")
''')


DEVICE_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    rng = np.random.default_rng(11)
    n = SC(3_000_001)
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    mask = pa.array(rng.random(n) < 0.1, mask=rng.random(n) < 0.02)
    idx = pa.array(rng.integers(0, n, SC(500_000)).astype(np.uint32), mask=rng.random(SC(500_000)) < 0.05)
    f64 = pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1)
    want = dict(f=pc.filter(vals, mask), fe=pc.filter(vals, mask, null_selection_behavior="emit_null"),
                fs=pc.filter(vals.slice(7), mask.slice(7)), t=pc.take(vals, idx),
                c=pc.cast(f64, pa.float32()), cs=pc.cast(f64.slice(9), pa.float32()))
    want_chain = pc.cast(pc.filter(f64, mask), pa.float32())
    # (pa.array(ndarray, type=...) itself calls `cast`: build these before the counters are read)
    temporal_keys = [pa.array(rng.integers(-2**62, 2**62, n), pa.timestamp("us"), mask=rng.random(n) < 0.02),
                     pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), pa.date32()),
                     pa.array(rng.integers(0, 86400, n).astype(np.int32), pa.time32("s"), mask=rng.random(n) < 0.02),
                     pa.array(rng.integers(-2**40, 2**40, n), pa.duration("ns"))]
    stock_before = sum(lib.arrow_amd_plugin_calls(f, 0) for f in (b"array_filter", b"array_take", b"cast"))

    d_vals, d_mask, d_idx, d_f64 = to_device(vals), to_device(mask), to_device(idx), to_device(f64)
    assert not d_vals.is_cpu and d_vals.device_type == pa.DeviceAllocationType.ROCM, d_vals.device_type
    got = dict(f=pc.filter(d_vals, d_mask), fe=pc.filter(d_vals, d_mask, null_selection_behavior="emit_null"),
               fs=pc.filter(d_vals.slice(7), d_mask.slice(7)), t=pc.take(d_vals, d_idx),
               c=pc.cast(d_f64, pa.float32()), cs=pc.cast(d_f64.slice(9), pa.float32()))
    for k, w in want.items():
        assert not got[k].is_cpu, ("output left the device", k)     # outputs stay in HBM
        h = to_host(got[k])               # (pyarrow refuses .null_count on non-CPU data)
        assert h.is_cpu, k
        if not h.equals(w):
            hv, wv = h.fill_null(0).to_numpy(zero_copy_only=False), w.fill_null(0).to_numpy(zero_copy_only=False)
            bad = np.nonzero((hv != wv) | (np.asarray(h.is_null()) != np.asarray(w.is_null())))[0] if len(h) == len(w) else []
            raise SystemExit(f"MISMATCH {k}: len {len(h)} vs {len(w)}, nulls {h.null_count} vs {w.null_count}, "
                             f"first bad {bad[:5]}, offset {h.offset}")
        assert h.null_count == w.null_count, (k, h.null_count, w.null_count)
    # compare on the device: bitmap + intersected validity stay in HBM and feed a filter directly
    f64b = pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.07)
    d_f64b = to_device(f64b)
    for dev_mask, host_mask in ((pc.greater(d_f64, d_f64b), pc.greater(f64, f64b)),
                                (pc.greater(d_f64.slice(5, n - 11), d_f64b.slice(11, n - 11)), pc.greater(f64.slice(5, n - 11), f64b.slice(11, n - 11))),
                                (pc.greater(d_f64, 0.25), pc.greater(f64, 0.25)),
                                (pc.greater(-0.5, d_f64b), pc.greater(-0.5, f64b))):
        assert not dev_mask.is_cpu
        hm = to_host(dev_mask)
        assert hm.equals(host_mask) and hm.null_count == host_mask.null_count
    sel = to_host(pc.filter(d_vals, pc.greater(d_f64, d_f64b)))
    assert sel.equals(pc.filter(vals, pc.greater(f64, f64b)))
    assert lib.arrow_amd_plugin_calls(b"greater", 1) >= 5
    # arithmetic and int64 compare on the device (wrap-around add, scalar operands)
    i64b = pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.05)
    d_i64b = to_device(i64b)
    for dev_out, host_out in ((pc.add(d_vals, d_i64b), pc.add(vals, i64b)), (pc.add(d_vals, 17), pc.add(vals, 17)),
                              (pc.add(-3, d_i64b), pc.add(-3, i64b)),
                              (pc.add(d_vals.slice(3, n - 9), d_i64b.slice(9, n - 9)), pc.add(vals.slice(3, n - 9), i64b.slice(9, n - 9))),
                              (pc.add(d_f64, d_f64b), pc.add(f64, f64b)), (pc.add(d_f64, 0.125), pc.add(f64, 0.125)),
                              (pc.greater(d_vals, d_i64b), pc.greater(vals, i64b)), (pc.greater(d_vals, 0), pc.greater(vals, 0)),
                              (pc.greater(12345, d_i64b), pc.greater(12345, i64b))):
        assert not dev_out.is_cpu
        ho = to_host(dev_out)
        assert ho.equals(host_out) and ho.null_count == host_out.null_count
    # the six device calls ran on the GPU; the six host references went to Arrow's stock kernel
    assert lib.arrow_amd_plugin_calls(b"add", 1) == 6 and lib.arrow_amd_plugin_calls(b"add", 0) == 6
    # compare -> filter -> add chain, all in HBM
    chain2 = to_host(pc.add(pc.filter(d_vals, pc.greater(d_vals, d_i64b)), 1))
    assert chain2.equals(pc.add(pc.filter(vals, pc.greater(vals, i64b)), 1))
    # subtract / multiply / *_checked on device arrays (what `-`, `*`, `+` on expressions mean); overflow is an error
    smalls = pa.array(rng.integers(-10**6, 10**6, n), mask=rng.random(n) < 0.05)
    smalls2 = pa.array(rng.integers(-10**6, 10**6, n), mask=rng.random(n) < 0.05)
    d_sm, d_sm2 = to_device(smalls), to_device(smalls2)
    for ar in (pc.subtract, pc.multiply, pc.add_checked, pc.subtract_checked, pc.multiply_checked):
        for dev_out, host_out in ((ar(d_sm, d_sm2), ar(smalls, smalls2)), (ar(d_sm, 3), ar(smalls, 3)), (ar(-7, d_sm2), ar(-7, smalls2)),
                                  (ar(d_f64, d_f64b), ar(f64, f64b)), (ar(d_sm.slice(5, n - 9), d_sm2.slice(9, n - 9)), ar(smalls.slice(5, n - 9), smalls2.slice(9, n - 9)))):
            assert not dev_out.is_cpu
            ho = to_host(dev_out)
            assert ho.equals(host_out) and ho.null_count == host_out.null_count, ar
    assert to_host(pc.subtract(d_vals, d_i64b)).equals(pc.subtract(vals, i64b))          # wrap-around
    for bad in (lambda: pc.add_checked(d_vals, d_i64b), lambda: pc.multiply_checked(d_vals, 4)):
        try:
            bad()
            raise SystemExit("expected overflow")
        except pa.lib.ArrowInvalid as e:
            assert str(e) == "overflow", str(e)
    # integer casts on device arrays: int64 -> int32 checked (first offending valid slot named) / unsafe, int32 -> int64
    casts_before = lib.arrow_amd_plugin_calls(b"cast", 1)
    host_reference_casts = 3          # pc.cast on HOST arrays below goes to Arrow's stock kernel
    i32ok = to_host(pc.cast(d_sm, pa.int32()))
    assert i32ok.equals(pc.cast(smalls, pa.int32()))
    assert to_host(pc.cast(pc.cast(d_sm.slice(7), pa.int32()), pa.int64())).equals(smalls.slice(7))
    try:
        pc.cast(d_vals, pa.int32())
        raise SystemExit("expected ArrowInvalid")
    except pa.lib.ArrowInvalid as e:
        try:
            pc.cast(vals, pa.int32())
        except pa.lib.ArrowInvalid as he:
            assert str(e) == str(he), (str(e), str(he))
    assert to_host(pc.cast(d_vals, pa.int32(), safe=False)).equals(pc.cast(vals, pa.int32(), safe=False))
    assert to_host(pc.cast(d_sm, pa.float64())).equals(pc.cast(smalls, pa.float64()))
    assert to_host(pc.cast(d_vals, pa.float64(), safe=False)).equals(pc.cast(vals, pa.float64(), safe=False))
    host_reference_casts += 2
    assert lib.arrow_amd_plugin_calls(b"cast", 1) == casts_before + 6     # (the failing call is not counted)
    # the whole comparison family on device arrays (NaN-aware for doubles), scalars on either side
    f64n = pa.array(np.where(rng.random(n) < 0.01, np.nan, np.round(rng.standard_normal(n) * 4) / 4), mask=rng.random(n) < 0.05)
    d_f64n = to_device(f64n)
    for cmp in (pc.equal, pc.not_equal, pc.greater_equal, pc.less, pc.less_equal):
        for dev_out, host_out in ((cmp(d_vals, d_i64b), cmp(vals, i64b)), (cmp(d_vals, 12345), cmp(vals, 12345)),
                                  (cmp(0.25, d_f64n), cmp(0.25, f64n)), (cmp(d_f64n, d_f64b), cmp(f64n, f64b)),
                                  (cmp(d_f64n.slice(9, n - 20), d_f64n.slice(3, n - 20)), cmp(f64n.slice(9, n - 20), f64n.slice(3, n - 20)))):
            assert not dev_out.is_cpu
            ho = to_host(dev_out)
            assert ho.equals(host_out) and ho.null_count == host_out.null_count, cmp
    assert lib.arrow_amd_plugin_calls(b"compare", 1) == 25
    # Kleene logic on device masks (what `&`, `|`, `~` on expressions mean) and a combined filter
    ma, mb = pc.greater(d_vals, d_i64b), pc.greater(d_f64, d_f64b)
    hma, hmb = pc.greater(vals, i64b), pc.greater(f64, f64b)
    for dev_out, host_out in ((pc.and_kleene(ma, mb), pc.and_kleene(hma, hmb)), (pc.or_kleene(ma, mb), pc.or_kleene(hma, hmb)),
                              (pc.invert(ma), pc.invert(hma)),
                              (pc.and_kleene(ma.slice(3, n - 70), mb.slice(70, n - 70)), pc.and_kleene(hma.slice(3, n - 70), hmb.slice(70, n - 70))),
                              (pc.or_kleene(d_mask, pc.invert(ma)), pc.or_kleene(mask, pc.invert(hma)))):
        assert not dev_out.is_cpu
        ho = to_host(dev_out)
        assert ho.equals(host_out) and ho.null_count == host_out.null_count
    assert lib.arrow_amd_plugin_calls(b"boolean", 1) == 6
    both = to_host(pc.filter(d_vals, pc.and_kleene(ma, pc.invert(mb))))
    assert both.equals(pc.filter(vals, pc.and_kleene(hma, pc.invert(hmb))))
    # sort on the device: uint64 indices stay in HBM and feed take
    skeys = pa.array(rng.integers(0, 2**63, n).astype(np.uint64), mask=rng.random(n) < 0.03)
    d_skeys = to_device(skeys)
    d_perm = pc.array_sort_indices(d_skeys, order="descending", null_placement="at_start")
    assert not d_perm.is_cpu
    assert to_host(d_perm).equals(pc.array_sort_indices(skeys, order="descending", null_placement="at_start"))
    assert to_host(pc.take(d_skeys, d_perm)).equals(pc.take(skeys, pc.array_sort_indices(skeys, order="descending", null_placement="at_start")))
    # temporal keys sort by their physical integers (timestamp/date64/duration/time64: int64; date32/time32: int32)
    light = os.environ.get("ARROW_AMD_TEST_LIGHT") == "1"      # the emulated run keeps one temporal type / order (sorts are slow there)
    for tkeys in (temporal_keys[:1] if light else temporal_keys):
        d_t = to_device(tkeys)
        for order, place in (("ascending", "at_end"), ("descending", "at_start"))[: 1 if light else 2]:
            got_p = pc.array_sort_indices(d_t, order=order, null_placement=place)
            assert not got_p.is_cpu
            assert to_host(got_p).equals(pc.array_sort_indices(tkeys, order=order, null_placement=place)), (tkeys.type, order)
    # a chain that never leaves the device: filter -> cast
    chain = to_host(pc.cast(pc.filter(d_f64, d_mask), pa.float32()))
    assert chain.equals(want_chain)
    # no device call above was handed to a stock kernel (host reference casts are: count them out)
    stock_now = {f: lib.arrow_amd_plugin_calls(f, 0) for f in (b"array_filter", b"array_take", b"cast")}
    assert sum(stock_now.values()) == stock_before + host_reference_casts, (stock_now, stock_before)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) >= 4 and lib.arrow_amd_plugin_calls(b"array_take", 1) >= 1
    # drop_null = Filter(values, <validity bitmap as a boolean array>) (vector_selection.cc:79-91): the
    # filter's data buffer IS the device validity buffer, so this is the device filter again
    # (only for arrays with a known null count: Array::null_count() on a SLICED device array would
    #  popcount HBM from the CPU inside Arrow itself)
    for dv, hv in ((d_vals, vals), (d_f64, f64)):
        got_d = pc.drop_null(dv)
        assert not got_d.is_cpu
        assert to_host(got_d).equals(pc.drop_null(hv))
    # the other fixed-width classes of match::Primitive() + decimal128 / decimal256 / fixed_size_binary (widths 2..32 bytes)
    import decimal
    nw = SC(200_003)
    wmask = pa.array(rng.random(nw) < 0.4, mask=rng.random(nw) < 0.03)
    widx = pa.array(rng.integers(0, nw, SC(50_000)).astype(np.int64), mask=rng.random(SC(50_000)) < 0.05)
    d_wmask, d_widx = to_device(wmask), to_device(widx)
    raw = rng.integers(-2**62, 2**62, nw)
    wides = [pa.array(raw, pa.timestamp("ns", tz="UTC"), mask=rng.random(nw) < 0.1),
             pa.array(raw, pa.duration("us")), pa.array(raw, pa.time64("ns")),
             pa.array((raw % 86400).astype(np.int32), pa.time32("s"), mask=rng.random(nw) < 0.1),
             pa.array((raw % 1000).astype(np.float16)),
             pa.Array.from_buffers(pa.decimal128(38, 4), nw // 2, [None, pa.py_buffer(raw[: nw // 2 * 2].tobytes())]),
             pa.Array.from_buffers(pa.binary(16), nw // 2, [None, pa.py_buffer(raw[: nw // 2 * 2].tobytes())]),
             # 32-byte values (decimal256, fixed_size_binary(32)): the widest case of PrimitiveFilterExec / FixedWidthTakeExec
             pa.Array.from_buffers(pa.decimal256(60, 4), nw // 4, [None, pa.py_buffer((raw[: nw // 4 * 4] % 10**15).tobytes())]),
             pa.Array.from_buffers(pa.binary(32), nw // 4, [pa.py_buffer(np.packbits(rng.random(nw // 4 + 8) > 0.1, bitorder="little").tobytes()),
                                                            pa.py_buffer(raw[: nw // 4 * 4].tobytes())]),
             pa.Array.from_buffers(pa.binary(2), nw, [None, pa.py_buffer(raw.astype(np.int16).tobytes())])]
    for hv in wides:
        dv = to_device(hv)
        mlen = len(hv)
        cases = [(pc.filter(dv, d_wmask.slice(0, mlen)), pc.filter(hv, wmask.slice(0, mlen))),
                 (pc.filter(dv, d_wmask.slice(0, mlen), null_selection_behavior="emit_null"),
                  pc.filter(hv, wmask.slice(0, mlen), null_selection_behavior="emit_null")),
                 (pc.take(dv, to_device(pc.min_element_wise(widx, mlen - 1))), pc.take(hv, pc.min_element_wise(widx, mlen - 1)))]
        for got_d, want_h in cases:
            assert not got_d.is_cpu, hv.type
            h = to_host(got_d)
            assert h.equals(want_h) and h.null_count == want_h.null_count, (hv.type, len(h), len(want_h))
    # utf8 / binary values in HBM: filter == take(GetTakeIndices) on the device, 3 buffers out
    ns = SC(400_003)
    lens = rng.integers(0, 20, ns)
    words = np.array(["".join(chr(97 + (i + j) % 26) for j in range(l)) for i, l in enumerate(lens[:5000])], dtype=object)
    strs = pa.array(np.tile(words, ns // 5000 + 1)[:ns], type=pa.string(), mask=rng.random(ns) < 0.1)
    smask = pa.array(rng.random(ns) < 0.3, mask=rng.random(ns) < 0.02)
    sidx = pa.array(rng.integers(0, ns, SC(100_000)).astype(np.int32), mask=rng.random(SC(100_000)) < 0.05)
    d_strs, d_smask, d_sidx = to_device(strs), to_device(smask), to_device(sidx)
    gpu_f, gpu_t = lib.arrow_amd_plugin_calls(b"array_filter", 1), lib.arrow_amd_plugin_calls(b"array_take", 1)
    for typ in (pa.string(), pa.binary()):
        hs = strs.cast(typ)
        ds = to_device(hs)
        cases = [(pc.filter(ds, d_smask), pc.filter(hs, smask)),
                 (pc.filter(ds, d_smask, null_selection_behavior="emit_null"), pc.filter(hs, smask, null_selection_behavior="emit_null")),
                 (pc.filter(ds.slice(13), d_smask.slice(13)), pc.filter(hs.slice(13), smask.slice(13))),
                 (pc.take(ds, d_sidx), pc.take(hs, sidx)),
                 (pc.take(ds.slice(5, 1000), to_device(pa.array([0, 999, 3], pa.int64()))), pc.take(hs.slice(5, 1000), pa.array([0, 999, 3])))]
        for got_d, want_h in cases:
            assert not got_d.is_cpu
            h = to_host(got_d)
            assert h.equals(want_h) and h.null_count == want_h.null_count, (typ, len(h), len(want_h))
    assert to_host(pc.drop_null(d_strs)).equals(pc.drop_null(strs))
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) == gpu_f + 7 and lib.arrow_amd_plugin_calls(b"array_take", 1) == gpu_t + 4
    try:
        pc.take(d_strs, to_device(pa.array(np.array([0, ns], dtype=np.int64))))
        raise SystemExit("expected IndexError")
    except pa.lib.ArrowIndexError as e:
        assert str(e) == f"Index {ns} out of bounds", str(e)
    try:
        pc.take(d_vals, to_device(pa.array(np.array([0, n], dtype=np.int64))))
        raise SystemExit("expected IndexError")
    except pa.lib.ArrowIndexError as e:
        assert str(e) == f"Index {n} out of bounds", str(e)
    print("DEVICE_OK")
''')


GOLDEN_SCRIPT = textwrap.dedent(r'''
    import ctypes, json, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))     # the golden arrays are tiny: send them to the GPU anyway
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(k) for k in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(k) for k in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    g0 = lib.arrow_amd_plugin_calls(b"array_sort_indices", 1)
    ran = 0
    emulated = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1"
    # (the SIMT emulator is single-threaded and slow: two key types, no concurrent consumes there)
    types = (pa.int64(), pa.float64()) if emulated else (pa.int64(), pa.uint64(), pa.int32(), pa.uint32(), pa.float64(), pa.float32())
    # SortTemporal / TemporalTypeParameters (vector_sort_test.cc:754-781, :913-951) assert the integral vectors on the temporal
    # types; the 8- and 16-bit integers sort on the device widened to 32 bits (host arrays of those keep the counting sort)
    temporal = (pa.date32(), pa.date64(), pa.timestamp("s"), pa.timestamp("ns", tz="America/Phoenix"), pa.duration("ms"),
                pa.time32("s"), pa.time32("ms"), pa.time64("us"), pa.time64("ns"))
    narrow = (pa.int8(), pa.uint8(), pa.int16(), pa.uint16())
    def physical(typ):
        return pa.int32() if typ in (pa.date32(), pa.time32("s"), pa.time32("ms")) else pa.int64()
    def cases_for(typ):
        if typ in temporal or typ in narrow:
            extra = gold["sort_indices_narrow_and_wide"].get(str(typ), [])
            base = gold["sort_indices_integral"]
            if emulated:      # (the SIMT emulator takes about a second per sort: the cases with nulls and ties only)
                base = [c for c in base if len(c["values"]) > 2 and (None in c["values"] or 50 in c["values"])]
            return base + extra
        extra = gold["sort_indices_narrow_and_wide"]["int64"] if typ == pa.int64() else []
        return gold["sort_indices_integral"] + gold["sort_indices_real"] + extra
    for typ in types + ((temporal[2], narrow[0], narrow[1]) if emulated else temporal + narrow):
        is_f = pa.types.is_floating(typ)
        for case in cases_for(typ):
            vals = case["values"]
            if not is_f and any(x == "NaN" or (isinstance(x, float) and x != int(x)) for x in vals if x is not None):
                continue
            if typ in temporal:
                arr = pa.array(vals, type=physical(typ)).cast(typ)
            else:
                arr = pa.array([None if x is None else (float("nan") if x == "NaN" else x) for x in vals], type=typ)
            # (emulated: the device route only — the host route is the same kernels behind an upload, and the kernel-level
            #  replay of tests/test_emu_parity.py runs every vector on every key type)
            for where in (("device",) if emulated and len(arr) else ("host", "device")):
                if where == "device" and len(arr) == 0:
                    continue
                a = to_device(arr) if where == "device" else arr
                # unmodified pyarrow.compute -> CallFunction -> the registered kernel
                got = pc.array_sort_indices(a, order=case["order"], null_placement=case["null_placement"])
                got = to_host(got) if where == "device" else got
                assert got.to_pylist() == case["want"], (str(typ), where, case)
                ran += 1
    # random 8- / 16-bit keys with nulls, sliced (bit offsets): the device route (widened to 32 bits) against the
    # reference's counting sort, which host arrays of these types keep — stable, so the indices are identical
    rng = np.random.default_rng(5)
    m = 3000 if emulated else 300_000
    for np_t, typ in ((np.int8, pa.int8()), (np.uint8, pa.uint8()), (np.int16, pa.int16()), (np.uint16, pa.uint16())):
        info = np.iinfo(np_t)
        host = pa.array(rng.integers(info.min, info.max, m, dtype=np_t, endpoint=True), typ, mask=rng.random(m) < 0.1)
        dev = to_device(host)
        for order, placement in ((("ascending", "at_end"),) if emulated else (("ascending", "at_end"), ("descending", "at_start"), ("descending", "at_end"))):
            s0 = lib.arrow_amd_plugin_calls(b"array_sort_indices", 0)
            want = pc.array_sort_indices(host.slice(5, m - 9), order=order, null_placement=placement)
            assert lib.arrow_amd_plugin_calls(b"array_sort_indices", 0) == s0 + 1, "host int8 / int16 keys keep the reference kernel"
            got = to_host(pc.array_sort_indices(dev.slice(5, m - 9), order=order, null_placement=placement))
            assert got.equals(want), (str(typ), order, placement)
            ran += 1
    used = lib.arrow_amd_plugin_calls(b"array_sort_indices", 1) - g0
    assert ran > (60 if emulated else 300) and used > (60 if emulated else 200), (ran, used)     # the cases really ran on the registered GPU kernel
    # SumOnly through Acero: the registered hash_sum(int64, uint32) vtable (GroupByNode) and the fused aggregate_rocm node
    s = gold["hash_sum_sum_only"]
    batches = [pa.record_batch({"argument": pa.array(b["argument"], pa.int64()), "key": pa.array(b["key"], pa.int32())})
               for b in s["batches"]]
    tab = pa.Table.from_batches(batches)
    for threads in ((False,) if emulated else (True, False)):
        r = tab.group_by("key", use_threads=threads).aggregate([("argument", "sum")]).sort_by("key")
        assert [[k, v] for k, v in zip(r.column("key").to_pylist(), r.column("argument_sum").to_pylist())] == s["want_sorted_by_key"]
        fused = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("argument", "hash_sum", None, "s")], keys=["key"])),
        ]).to_table(use_threads=threads).sort_by("key")
        assert [[k, v] for k, v in zip(fused.column("key").to_pylist(), fused.column("s").to_pylist())] == s["want_sorted_by_key"]
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) > 0
    print("GOLDEN_OK", ran, used)
''')


ACERO_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    rng = np.random.default_rng(21)

    def fused(table, opts=None, name="v_sum"):
        decl = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", opts, name)], keys=["k"])),
        ])
        return decl.to_table()

    def same(got, want):
        got, want = got.sort_by("k"), want.select(["k", "v_sum"]).sort_by("k")
        assert got.schema.names == ["k", "v_sum"], got.schema
        assert got.equals(want), (got.slice(0, 5), want.slice(0, 5))

    n = SC(3_000_000)
    # no nulls: every batch is staged on the device, one radix-partitioned consume at the end
    t = pa.table({"x": pa.array(rng.random(n)), "k": pa.array(rng.integers(-70000, 70000, n).astype(np.int32)),
                  "v": pa.array(rng.integers(-2**63, 2**63 - 1, n))})
    same(fused(t), t.group_by("k", use_threads=False).aggregate([("v", "sum")]))
    # nulls in keys and values: consumed batch by batch; options honoured
    tn = pa.table({"k": pa.array(rng.integers(-300, 300, n).astype(np.int32), mask=rng.random(n) < 0.01),
                   "v": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.2)})
    same(fused(tn), tn.group_by("k", use_threads=False).aggregate([("v", "sum")]))
    o = pc.ScalarAggregateOptions(skip_nulls=False, min_count=3)
    same(fused(tn, o), tn.group_by("k", use_threads=False).aggregate([("v", "sum", o)]))
    # mixed: some chunks with nulls, some without, several chunks
    tm = pa.concat_tables([t.select(["k", "v"]).slice(0, n // 6), tn.slice(0, n // 7), t.select(["k", "v"]).slice(n // 6, n // 4)])
    same(fused(tm), tm.group_by("k", use_threads=False).aggregate([("v", "sum")]))
    # several aggregates over the same value column share one fused pass: sum, count (valid values),
    # a second sum with other options; host and device-resident input
    o3 = pc.ScalarAggregateOptions(skip_nulls=False, min_count=3)
    multi = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(tn)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions(
            [("v", "hash_sum", None, "s"), ("v", "hash_count", None, "c"), ("v", "hash_sum", o3, "s3")], keys=["k"])),
    ]).to_table().sort_by("k")
    ref = tn.group_by("k", use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "sum", o3)]).sort_by("k")
    assert multi.schema.names == ["k", "s", "c", "s3"], multi.schema
    assert multi.column("k").equals(ref.column("k"))
    for ours, theirs in (("s", 1), ("c", 2), ("s3", 3)):     # (pyarrow puts the key column first)
        assert multi.column(ours).equals(ref.column(theirs)), ours
    assert multi.column("c").null_count == 0
    # hash_min / hash_max (one more pass over the same fused table), mixed with sum and count; the
    # batches with nulls are consumed on arrival, so the table is rehashed (export -> merge) on the way
    for tab in (tn, tm, t.select(["k", "v"])):
        mm = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions(
                [("v", "hash_min", None, "lo"), ("v", "hash_sum", None, "s"), ("v", "hash_max", o3, "hi3"),
                 ("v", "hash_count", None, "c"), ("v", "hash_max", None, "hi")], keys=["k"])),
        ]).to_table().sort_by("k")
        ref = tab.group_by("k", use_threads=False).aggregate(
            [("v", "min"), ("v", "sum"), ("v", "max", o3), ("v", "count"), ("v", "max")]).sort_by("k")
        assert mm.schema.names == ["k", "lo", "s", "hi3", "c", "hi"], mm.schema
        assert mm.column("k").equals(ref.column("k"))
        for i, name in enumerate(["lo", "s", "hi3", "c", "hi"]):
            assert mm.column(name).equals(ref.column(1 + i)), (name, mm.column(name).slice(0, 5), ref.column(1 + i).slice(0, 5))
    # hash_mean(int64): float64 column, bit-equal to the reference's row-order double accumulation while the partial
    # sums stay exact integers (values below 2^31 here); declined with the reason for full-range values
    tsmall = pa.table({"k": tn.column("k"), "v": pa.array(rng.integers(-2**31, 2**31, n), mask=rng.random(n) < 0.2)})
    for o_mean in (None, pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)):
        mean = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tsmall)),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions(
                [("v", "hash_mean", o_mean, "m"), ("v", "hash_sum", None, "s")], keys=["k"])),
        ]).to_table().sort_by("k")
        ref = tsmall.group_by("k", use_threads=False).aggregate([("v", "mean", o_mean), ("v", "sum")]).sort_by("k")
        assert mean.schema.field("m").type == pa.float64()
        assert mean.column("k").equals(ref.column("k")) and mean.column("s").equals(ref.column(2))
        assert mean.column("m").equals(ref.column(1)), (mean.column("m").slice(0, 5), ref.column(1).slice(0, 5))
    try:
        acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tn)),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_mean", None, "m")], keys=["k"]))]).to_table()
        raise SystemExit("expected NotImplemented for hash_mean over full-range int64")
    except pa.lib.ArrowNotImplementedError as e:
        assert "2^53" in str(e)
    # what the fused int32 -> int64 operator refuses goes to the Grouper-based node of the same factory (round 3):
    # CountOptions(mode="all"), a float64 key (keys compare by their bits, as in the reference's row encoding)
    call = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(tn)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_count", pc.CountOptions(mode="all"), "c")], keys=["k"]))]).to_table().sort_by("k")
    ref = tn.group_by("k", use_threads=False).aggregate([("v", "count", pc.CountOptions(mode="all"))]).sort_by("k")
    assert call.column("k").equals(ref.column("k")) and call.column("c").equals(ref.column("v_count"))
    empty = pa.table({"k": pa.array([], pa.int32()), "v": pa.array([], pa.int64())})
    assert fused(empty).num_rows == 0
    assert fused(pa.table({"k": pa.array([1.0, 2.0, 1.0]), "v": pa.array([1, 2, 3], pa.int64())})).sort_by("k").column("v_sum").to_pylist() == [4, 2]
    # a utf8 key goes to the Grouper-based node too (round 3, commit 9c16ec1): equal to the stock GroupByNode, incl. the
    # empty string vs null, NUL bytes and shared prefixes
    ts = pa.table({"k": pa.array(["a", "", None, "a\x00", "a", "ab", None, "", "a\x00", "abcdefghijklmnopqrstuvwxyz"]),
                   "v": pa.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], pa.int64())})
    got = fused(ts).sort_by("k")
    ref = ts.group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k")
    assert got.column("k").equals(ref.column("k")) and got.column("v_sum").equals(ref.column("v_sum")), (got, ref)
    # what neither node takes is refused with the reason: a list key
    try:
        fused(pa.table({"k": pa.array([[1], [2]], pa.list_(pa.int32())), "v": pa.array([1, 2], pa.int64())}))
        raise SystemExit("expected NotImplemented for a list key")
    except pa.lib.ArrowNotImplementedError as e:
        assert "key column" in str(e), e
    print("ACERO_OK")
''')


ACERO_DEVICE_SCRIPT = textwrap.dedent(r'''
    import ctypes, faulthandler, os, sys
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    rng = np.random.default_rng(5)
    n = SC(1_000_003)
    for null_p in (0.0, 0.03):
        mk = (lambda a: pa.array(a, mask=rng.random(n) < null_p)) if null_p else pa.array
        k = mk(rng.integers(-5000, 5000, n).astype(np.int32))
        v = mk(rng.integers(-2**62, 2**62, n))          # (checked arithmetic in the projection: keep away from the edges)
        w = mk(rng.integers(-100, 100, n))
        host = pa.table({"k": k, "v": v, "w": w})
        dev = pa.table({"k": to_device(k), "v": to_device(v), "w": to_device(w)})

        def plan(table, agg):
            return acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
                acero.Declaration("filter", acero.FilterNodeOptions((pc.field("w") > 10) & ~(pc.field("v") > 2**61))),
                acero.Declaration("project", acero.ProjectNodeOptions([pc.field("k"), pc.field("v") - pc.field("w") * 2], ["k", "v"])),   # (`-`, `*` on expressions: subtract_checked, multiply_checked)
                acero.Declaration(agg, acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"])),
            ])

        want = plan(host, "aggregate").to_table(use_threads=False).select(["k", "v_sum"]).sort_by("k")
        names = (b"greater", b"add", b"array_filter", b"hash_sum", b"boolean")
        gpu0 = {f: lib.arrow_amd_plugin_calls(f, 1) for f in names}
        stock0 = {f: lib.arrow_amd_plugin_calls(f, 0) for f in names}
        lib.arrow_amd_plugin_aggregate_flushes.restype = ctypes.c_int64
        lib.arrow_amd_plugin_set_aggregate_flush_rows.argtypes = [ctypes.c_int64]
        # (round 6: a stock table_source over a device table delivers whole chunks by default; this case is about the node's
        #  staging of MANY small batches, so the reference SourceNode's 32Ki-row morsels are asked for: the opt-out)
        assert lib.arrow_amd_override_acero_factories(-1) == 0, lib.arrow_amd_plugin_last_error()
        for threads in (False, True):
            # null-free device batches are remembered and copied into the node's staging columns many at a time by
            # one launch (arx_copy_segments): with the default threshold (one copy at the end) and with a small one
            # (several copies while batches still arrive)
            for flush_rows in (1 << 21, max(1000, n // 40)):
                lib.arrow_amd_plugin_set_aggregate_flush_rows(flush_rows)
                f0 = lib.arrow_amd_plugin_aggregate_flushes()
                got = plan(dev, "aggregate_rocm").to_table(use_threads=threads).sort_by("k")
                assert got.schema.names == ["k", "v_sum"]
                assert got.equals(want), (null_p, threads, flush_rows, got.slice(0, 5), want.slice(0, 5))
                # (batches WITH nulls are staged the same way, their validity by arx_bitmap_copy_segments)
                flushes = lib.arrow_amd_plugin_aggregate_flushes() - f0
                assert flushes >= 1 and (flush_rows >= n or flushes > 1), (null_p, flush_rows, flushes)
            if null_p:   # the route before: every batch with nulls consumed on its own
                lib.arrow_amd_plugin_set_aggregate_stage_nulls(0)
                f0 = lib.arrow_amd_plugin_aggregate_flushes()
                got = plan(dev, "aggregate_rocm").to_table(use_threads=threads).sort_by("k")
                assert got.equals(want) and lib.arrow_amd_plugin_aggregate_flushes() == f0
                lib.arrow_amd_plugin_set_aggregate_stage_nulls(1)
        lib.arrow_amd_plugin_set_aggregate_flush_rows(1 << 21)
        assert lib.arrow_amd_override_acero_factories(0) == 0
        for f in names:   # FilterNode's expression, its per-column Filter, the projection and the group-by all ran on the GPU
            assert lib.arrow_amd_plugin_calls(f, 1) > gpu0[f], f
            assert lib.arrow_amd_plugin_calls(f, 0) == stock0[f], f

    # a table whose first chunk has no validity at all and whose second has nulls: the staged validity starts in the
    # middle of the plan (the rows staged before are marked valid afterwards), with flushes before and after the switch
    m = SC(300_000)
    ka, va = pa.array(rng.integers(-300, 300, m).astype(np.int32)), pa.array(rng.integers(-2**40, 2**40, m))
    kb = pa.array(rng.integers(-300, 300, m).astype(np.int32), mask=rng.random(m) < 0.1)
    vb = pa.array(rng.integers(-2**40, 2**40, m), mask=rng.random(m) < 0.2)
    host2 = pa.table({"k": pa.chunked_array([ka, kb]), "v": pa.chunked_array([va, vb])})
    dev2 = pa.table({"k": pa.chunked_array([to_device(ka), to_device(kb)]), "v": pa.chunked_array([to_device(va), to_device(vb)])})

    def plan2(table, agg):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
            acero.Declaration(agg, acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum"), ("v", "hash_count", None, "v_n")], keys=["k"])),
        ])

    want2 = plan2(host2, "aggregate").to_table(use_threads=False).select(["k", "v_sum", "v_n"]).sort_by("k")
    for flush_rows in (1 << 21, max(1000, m // 7)):
        lib.arrow_amd_plugin_set_aggregate_flush_rows(flush_rows)
        got2 = plan2(dev2, "aggregate_rocm").to_table(use_threads=False).select(["k", "v_sum", "v_n"]).sort_by("k")
        assert got2.equals(want2), (flush_rows, got2.slice(0, 5), want2.slice(0, 5))
    lib.arrow_amd_plugin_set_aggregate_flush_rows(1 << 21)

    # device arrays that start in the middle of their buffers (array offsets that are no multiple of 8: the staged
    # validity ranges start at any bit).  Added after the round's last GPU-box run, so for now on the emulated tier only.
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        for start, length in ((13, 2 * m - 40), (1, m + 1), (m // 3 + 5, m), (7, 9)):
            hk, hv = host2.column("k").combine_chunks().slice(start, length), host2.column("v").combine_chunks().slice(start, length)
            dk, dv = to_device(hk), to_device(hv)
            assert dk.offset == start
            want3 = plan2(pa.table({"k": hk, "v": hv}), "aggregate").to_table(use_threads=False).select(["k", "v_sum", "v_n"]).sort_by("k")
            for flush_rows in (1 << 21, max(1000, m // 7)):
                lib.arrow_amd_plugin_set_aggregate_flush_rows(flush_rows)
                got3 = plan2(pa.table({"k": dk, "v": dv}), "aggregate_rocm").to_table(use_threads=False).select(["k", "v_sum", "v_n"]).sort_by("k")
                assert got3.equals(want3), (start, length, flush_rows)
        lib.arrow_amd_plugin_set_aggregate_flush_rows(1 << 21)

    print("ACERO_DEVICE_OK")
''')


BOOLEAN_VALUES_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    rng = np.random.default_rng(29)
    n = SC(3_000_001)
    mask = pa.array(rng.random(n) < 0.1, mask=rng.random(n) < 0.02)
    idx = pa.array(rng.integers(0, n, SC(500_000)).astype(np.uint32), mask=rng.random(SC(500_000)) < 0.05)
    d_mask, d_idx = to_device(mask), to_device(idx)
    gpu_f, gpu_t = lib.arrow_amd_plugin_calls(b"array_filter", 1), lib.arrow_amd_plugin_calls(b"array_take", 1)
    # BOOLEAN (bit-packed) values on the device: the 1-bit gather
    bvals = pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.07)
    d_bvals = to_device(bvals)
    for got_d, want_h in ((pc.filter(d_bvals, d_mask), pc.filter(bvals, mask)),
                          (pc.filter(d_bvals.slice(9), d_mask.slice(9), null_selection_behavior="emit_null"),
                           pc.filter(bvals.slice(9), mask.slice(9), null_selection_behavior="emit_null")),
                          (pc.take(d_bvals, d_idx), pc.take(bvals, idx))):
        assert not got_d.is_cpu
        hb = to_host(got_d)
        assert hb.equals(want_h) and hb.null_count == want_h.null_count
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) == gpu_f + 2 and lib.arrow_amd_plugin_calls(b"array_take", 1) == gpu_t + 1
    print("BOOLEAN_VALUES_OK")
''')


MORSEL_FILTER_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    from pyarrow import acero
    lib.arrow_amd_plugin_set_filter_morsel_rows.argtypes = [ctypes.c_int64]
    rng = np.random.default_rng(31)
    # single-synchronisation filter path (arrow_amd_plugin_set_filter_morsel_rows): same results as the default path
    for n in (1, 63, 64, 65, 1000, 32768, SC(1_000_003)):
        for vnull, mnull, true_p in ((0.0, 0.0, 0.1), (0.1, 0.0, 0.5), (0.1, 0.05, 0.3), (1.0, 0.5, 1.0), (0.0, 0.0, 0.0)):
            mk = lambda a, p: pa.array(a, mask=rng.random(n) < p) if p else pa.array(a)
            cols = [mk(rng.integers(-2**62, 2**62, n), vnull), mk(rng.integers(-100, 100, n).astype(np.int32), vnull),
                    mk(rng.integers(0, 200, n).astype(np.uint8), vnull), mk(rng.standard_normal(n), vnull)]
            mask = mk(rng.random(n) < true_p, mnull)
            d_mask = to_device(mask)
            for col in cols:
                d_col = to_device(col)
                for sel in ("drop", "emit_null"):
                    want = pc.filter(col, mask, null_selection_behavior=sel)
                    lib.arrow_amd_plugin_set_filter_morsel_rows(0)
                    base = to_host(pc.filter(d_col, d_mask, null_selection_behavior=sel))
                    lib.arrow_amd_plugin_set_filter_morsel_rows(1 << 21)
                    got_d = pc.filter(d_col, d_mask, null_selection_behavior=sel)
                    assert not got_d.is_cpu
                    got = to_host(got_d)
                    assert got.equals(want) and got.equals(base) and got.null_count == want.null_count, (n, vnull, mnull, true_p, col.type, sel)
                    if n > 100:      # sliced operands
                        got = to_host(pc.filter(d_col.slice(7, n - 20), d_mask.slice(13, n - 20), null_selection_behavior=sel))
                        assert got.equals(pc.filter(col.slice(7, n - 20), mask.slice(13, n - 20), null_selection_behavior=sel))
    # an Acero plan over a device table: every FilterNode batch is a morsel
    m = SC(400_003)
    k, v, w = (pa.array(rng.integers(-500, 500, m).astype(np.int32)), pa.array(rng.integers(-2**40, 2**40, m), mask=rng.random(m) < 0.05),
               pa.array(rng.integers(-100, 100, m)))
    host = pa.table({"k": k, "v": v, "w": w})
    dev = pa.table({"k": to_device(k), "v": to_device(v), "w": to_device(w)})
    def plan(t, agg):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
            acero.Declaration("filter", acero.FilterNodeOptions(pc.field("w") > 10)),
            acero.Declaration(agg, acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))])
    want = plan(host, "aggregate").to_table(use_threads=False).select(["k", "v_sum"]).sort_by("k")
    for threads in (False, True):
        assert plan(dev, "aggregate_rocm").to_table(use_threads=threads).sort_by("k").equals(want)
    lib.arrow_amd_plugin_set_filter_morsel_rows(0)
    print("MORSEL_FILTER_OK")
''')


SELECTION_META_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    rng = np.random.default_rng(37)
    n = SC(1_000_003)
    mk = lambda a, p=0.05: pa.array(a, mask=rng.random(len(a)) < p)
    cols = {"i64": mk(rng.integers(-2**62, 2**62, n)), "i32": mk(rng.integers(-100, 100, n).astype(np.int32)),
            "f64": pa.array(rng.standard_normal(n)), "flag": mk(rng.random(n) < 0.5),
            "s": pa.array(np.array(["", "a", "bb", "gfx950", "MI355X"], dtype=object)[rng.integers(0, 5, n)], type=pa.string(), mask=rng.random(n) < 0.1),
            "ts": pa.array(rng.integers(0, 2**50, n), pa.timestamp("us"))}
    mask = mk(rng.random(n) < 0.2, 0.03)
    idx = mk(rng.integers(0, n, SC(200_000)), 0.02)
    d_cols = {k: to_device(v) for k, v in cols.items()}
    d_mask, d_idx = to_device(mask), to_device(idx)

    def host_table(t):
        return pa.table({name: pa.chunked_array([c if c.is_cpu else to_host(c) for c in t.column(name).chunks], t.schema.field(name).type)
                         for name in t.schema.names})

    # FilterMetaFunction / TakeMetaFunction shapes over device-resident data: record batch, table, chunked array.
    # By NAME, as CallFunction / Acero / any C++ caller does: pyarrow's generated wrappers (pc.filter, pc.take) hold the
    # function objects they found when pyarrow.compute was imported, so they see a re-registered meta-function only if
    # the plugin was loaded first (and Table.filter / Table.take refuse non-CPU data on their own).
    def dev_filter(values, selection, sel="drop"):
        return pc.call_function("filter", [values, selection], pc.FilterOptions(null_selection_behavior=sel))

    def dev_take(values, indices):
        return pc.call_function("take", [values, indices])

    h_batch, d_batch = pa.record_batch(cols), pa.record_batch(d_cols)
    h_table, d_table = pa.table(cols), pa.table(d_cols)
    f0, t0 = lib.arrow_amd_plugin_calls(b"array_filter", 1), lib.arrow_amd_plugin_calls(b"array_take", 1)
    tc0 = lib.arrow_amd_plugin_calls(b"take_columns", 1)
    for sel in ("drop", "emit_null"):
        want = pc.filter(h_table, mask, null_selection_behavior=sel)
        got_b = dev_filter(d_batch, d_mask, sel)
        assert isinstance(got_b, pa.RecordBatch) and not got_b.column(0).is_cpu
        assert host_table(pa.Table.from_batches([got_b])).equals(want), sel
        got_t = dev_filter(d_table, d_mask, sel)
        assert isinstance(got_t, pa.Table) and host_table(got_t).equals(want), sel
        got_c = dev_filter(d_table.column("i64"), d_mask, sel)
        assert isinstance(got_c, pa.ChunkedArray) and to_host(got_c.chunk(0)).equals(want.column("i64").combine_chunks())
    want = pc.take(h_table, idx)
    assert host_table(pa.Table.from_batches([dev_take(d_batch, d_idx)])).equals(want)
    assert host_table(dev_take(d_table, d_idx)).equals(want)
    assert to_host(dev_take(d_table.column("s"), d_idx).chunk(0)).equals(want.column("s").combine_chunks())
    ncols = len(cols)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) == f0 + 2 * (2 * ncols + 1)
    assert lib.arrow_amd_plugin_calls(b"array_take", 1) >= t0 + 2 * ncols + 1
    # the four fixed-width columns of the batch and of the table went through ONE launch each (arx_take_columns)
    assert lib.arrow_amd_plugin_calls(b"take_columns", 1) == tc0 + 2
    # bounds errors keep the reference's message
    try:
        dev_take(d_table, to_device(pa.array([0, n], pa.int64())))
        raise SystemExit("expected IndexError")
    except pa.lib.ArrowIndexError as e:
        assert str(e) == f"Index {n} out of bounds", str(e)
    # a device column in several chunks is refused (Concatenate runs on the CPU), not crashed on
    two = pa.chunked_array([to_device(cols["i64"].slice(0, 1000)), to_device(cols["i64"].slice(1000, 1000))])
    try:
        dev_filter(two, to_device(mask.slice(0, 2000)))
        raise SystemExit("expected NotImplemented")
    except pa.lib.ArrowNotImplementedError as e:
        assert "chunks" in str(e), str(e)
    # sort_indices of a device table / record batch / chunked array: several keys, per-key direction and null placement;
    # the indices stay in HBM and feed take (= an ORDER BY expressed as two function calls)
    sk = [("i32", "ascending", "at_start"), ("ts", "descending", "at_end"), ("i64", "ascending", "at_end")]
    for keys in (sk, sk[:1], sk[1:2]):
        want_idx = pc.sort_indices(h_table, sort_keys=keys)
        got_idx = pc.call_function("sort_indices", [d_table], pc.SortOptions(sort_keys=keys))
        assert not got_idx.is_cpu and to_host(got_idx).equals(want_idx), keys
        assert to_host(pc.call_function("sort_indices", [d_batch], pc.SortOptions(sort_keys=keys))).equals(want_idx), keys
    assert host_table(dev_take(d_table, got_idx)).equals(pc.take(h_table, want_idx))
    assert to_host(pc.call_function("sort_indices", [d_table.column("f64")], pc.SortOptions(sort_keys=[("", "descending")]))).equals(
        pc.sort_indices(h_table.column("f64"), sort_keys=[("", "descending")]))
    # (round 5) a device column in SEVERAL chunks is concatenated in HBM first: indices into the logical column, as the
    # reference's ChunkedArray sorters give them; also as the key column of a table in several chunks
    cuts = [0, n // 3, n // 3 + n // 5, n]
    host_chunked = pa.chunked_array([cols["f64"].slice(a, b - a) for a, b in zip(cuts, cuts[1:])])
    dev_chunked = pa.chunked_array([to_device(c) for c in host_chunked.chunks])
    assert to_host(pc.call_function("sort_indices", [dev_chunked], pc.SortOptions(sort_keys=[("", "descending")]))).equals(
        pc.sort_indices(host_chunked, sort_keys=[("", "descending")]))
    h3 = pa.Table.from_batches([h_table.slice(a, b - a).to_batches()[0] for a, b in zip(cuts, cuts[1:])])
    d3 = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c) for c in rb.columns], names=rb.schema.names) for rb in h3.select(["i32", "ts", "i64"]).to_batches()])
    assert to_host(pc.call_function("sort_indices", [d3], pc.SortOptions(sort_keys=sk))).equals(pc.sort_indices(h3, sort_keys=sk))
    # (round 5) utf8 / binary sort keys: the string as (8-byte big-endian chunks, length) keys of the same chain
    for keys in ([("s", "ascending")], [("s", "descending"), ("i32", "ascending")], [("i32", "descending", "at_start"), ("s", "ascending", "at_start")]):
        want_idx = pc.sort_indices(h_table, sort_keys=keys)
        got_idx = pc.call_function("sort_indices", [d_table], pc.SortOptions(sort_keys=keys))
        assert to_host(got_idx).equals(want_idx), keys
    assert pc.sort_indices(h_table, sort_keys=sk).equals(pc.call_function("sort_indices", [h_table], pc.SortOptions(sort_keys=sk)))   # host: stock
    # host data: the stock meta-functions, untouched
    assert pc.filter(h_table, mask).equals(h_table.filter(mask)) and pc.take(h_batch, idx).equals(pa.record_batch(cols).take(idx))
    assert pc.filter(pa.chunked_array([cols["i64"].slice(0, 1000), cols["i64"].slice(1000, 1000)]), mask.slice(0, 2000)).length() > 0
    # casts the device path does not cover are refused, not handed to a CPU kernel (numeric pairs are all covered since
    # round 3: VECTOR_HASH_SCRIPT); same-type casts are zero-copy
    assert to_host(pc.cast(d_cols["i32"], pa.float32(), safe=False)).equals(pc.cast(cols["i32"], pa.float32(), safe=False))
    try:
        pc.cast(d_cols["i32"], pa.string())
        raise SystemExit("expected NotImplemented")
    except pa.lib.ArrowNotImplementedError as e:
        assert "device-resident" in str(e), str(e)
    assert to_host(pc.cast(d_cols["i64"], pa.int64())).equals(cols["i64"])
    print("SELECTION_META_OK")
''')


DIVIDE_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    from pyarrow import acero
    rng = np.random.default_rng(41)
    n = SC(2_000_003)
    MIN = -2**63
    mk = lambda a, p=0.08: pa.array(a, mask=rng.random(len(a)) < p)
    li = mk(rng.integers(-2**62, 2**62, n)); ri_raw = rng.integers(-60, 60, n); ri_raw[ri_raw == 0] = 3
    ri = mk(ri_raw)
    lf = mk(np.round(rng.standard_normal(n) * 8) / 4); rf = mk(np.where(rng.random(n) < 0.2, 0.0, np.round(rng.standard_normal(n) * 4) / 2))
    d_li, d_ri, d_lf, d_rf = to_device(li), to_device(ri), to_device(lf), to_device(rf)
    g0 = lib.arrow_amd_plugin_calls(b"add", 1)
    for fn in (pc.divide, pc.divide_checked):
        for dev_out, host_out in ((fn(d_li, d_ri), fn(li, ri)), (fn(d_li, 7), fn(li, 7)), (fn(-1000003, d_ri), fn(-1000003, ri)),
                                  (fn(d_li.slice(5, n - 9), d_ri.slice(9, n - 9)), fn(li.slice(5, n - 9), ri.slice(9, n - 9)))):
            assert not dev_out.is_cpu
            ho = to_host(dev_out)
            assert ho.equals(host_out) and ho.null_count == host_out.null_count, fn
    # doubles: IEEE division (inf / nan where the divisor is 0) unchecked; compare bit patterns
    got, want = to_host(pc.divide(d_lf, d_rf)), pc.divide(lf, rf)
    assert np.array_equal(np.asarray(got.is_null()), np.asarray(want.is_null()))
    assert np.array_equal(pc.fill_null(got, 0.0).to_numpy().view(np.uint64), pc.fill_null(want, 0.0).to_numpy().view(np.uint64))
    assert lib.arrow_amd_plugin_calls(b"add", 1) == g0 + 9
    # errors: the LAST failing valid slot names the Status; failing values under nulls do not fail
    def message(call):
        try:
            call()
            return None
        except pa.lib.ArrowInvalid as e:
            return str(e)
    zi = pa.array([5, MIN, 7, 1, 9], pa.int64()); zd = pa.array([1, -1, 0, 2, 3], pa.int64())
    cases = [(zi, zd), (pa.array([5, 0, MIN]), pa.array([0, 1, -1])), (pa.array([MIN, 4]), pa.array([-1, 2])),
             (pa.array([5, None, MIN]), pa.array([None, 0, 1])), (pa.array([1, 2]), pa.array([0, None]))]
    for l, r in cases:
        for fn in (pc.divide, pc.divide_checked):
            want = message(lambda: fn(l, r))
            got = message(lambda: fn(to_device(l), to_device(r)))
            assert got == want, (l, r, fn, got, want)
            if want is None:
                assert to_host(fn(to_device(l), to_device(r))).equals(fn(l, r))
    assert message(lambda: pc.divide(d_li, 0)) == "divide by zero" == message(lambda: pc.divide(li, 0))
    assert message(lambda: pc.divide_checked(d_lf, d_rf)) == "divide by zero" == message(lambda: pc.divide_checked(lf, rf))
    # `/` on an Acero expression means divide_checked? no: pyarrow maps it to `divide`; either way it stays on the device
    t_host = pa.table({"a": li, "b": ri}); t_dev = pa.table({"a": d_li, "b": d_ri})
    def plan(t):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
            acero.Declaration("project", acero.ProjectNodeOptions([pc.field("a") / pc.field("b")], ["q"])),
            acero.Declaration("aggregate", acero.AggregateNodeOptions([("q", "sum", None, "s"), ("q", "count", None, "c")]))])
    assert plan(t_dev).to_table(use_threads=False).equals(plan(t_host).to_table(use_threads=False))
    print("DIVIDE_OK")
''')


NUMERIC_OPS_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    from pyarrow import acero
    rng = np.random.default_rng(43)
    n = SC(1_000_003)
    cmp_fns = ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal")
    ari_fns = ("add", "subtract", "multiply", "add_checked", "subtract_checked", "multiply_checked")
    types = [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.uint64(), pa.float32()]
    c0, a0 = lib.arrow_amd_plugin_calls(b"compare", 1) + lib.arrow_amd_plugin_calls(b"greater", 1), lib.arrow_amd_plugin_calls(b"add", 1)
    ran_c = ran_a = 0
    for t in types:
        npdt = t.to_pandas_dtype()
        if pa.types.is_floating(t):
            la, lb = (np.round(rng.standard_normal(n) * 4) / 2).astype(npdt), (np.round(rng.standard_normal(n) * 4) / 2).astype(npdt)
            sc = 0.5
        else:
            hi = 11        # sums, differences (signed) and products of two values stay inside every type
            lo = -hi if pa.types.is_signed_integer(t) else 0
            la, lb = rng.integers(lo, hi + 1, n).astype(npdt), rng.integers(lo, hi + 1, n).astype(npdt)
            if not pa.types.is_signed_integer(t):
                la = (la + hi).astype(npdt)      # a >= b: unsigned differences do not wrap / overflow
            sc = 3
        a = pa.array(la, t, mask=rng.random(n) < 0.08)
        b = pa.array(lb, t, mask=rng.random(n) < 0.05)
        da, db = to_device(a), to_device(b)
        s = pa.scalar(sc, t)
        for name in cmp_fns + ari_fns:
            fn = lambda x, y: pc.call_function(name, [x, y])
            for dev_out, host_out in ((fn(da, db), fn(a, b)), (fn(da, s), fn(a, s)),
                                      (fn(da.slice(5, n - 9), db.slice(9, n - 9)), fn(a.slice(5, n - 9), b.slice(9, n - 9)))):
                assert not dev_out.is_cpu, (name, t)
                ho = to_host(dev_out)
                assert ho.type == host_out.type and ho.equals(host_out) and ho.null_count == host_out.null_count, (name, t)
            if name in cmp_fns: ran_c += 3
            else: ran_a += 3
        # the type's own overflow: unchecked wraps in ITS width, checked fails with the reference's text — unless the slot is null
        if pa.types.is_integer(t):
            info = np.iinfo(npdt)
            edge, one = pa.array([info.max, 5, info.min], t), pa.array([1, 2, 0], t)
            assert to_host(pc.add(to_device(edge), to_device(one))).equals(pc.add(edge, one))
            ran_a += 1
            for l, r in ((edge, one), (pa.array([info.max, 5, None], t), pa.array([None, 2, 1], t))):
                try:
                    want = pc.add_checked(l, r)
                except pa.lib.ArrowInvalid as e:
                    want = str(e)
                try:
                    got = to_host(pc.add_checked(to_device(l), to_device(r)))
                    ran_a += 1
                except pa.lib.ArrowInvalid as e:
                    got = str(e)
                assert (got == want) if isinstance(want, str) else got.equals(want), (t, got, want)
    c1, a1 = lib.arrow_amd_plugin_calls(b"compare", 1) + lib.arrow_amd_plugin_calls(b"greater", 1), lib.arrow_amd_plugin_calls(b"add", 1)
    assert c1 - c0 == ran_c and a1 - a0 == ran_a, (c1 - c0, ran_c, a1 - a0, ran_a)
    # temporal types: the comparisons of timestamp / duration / time32 / time64 (per unit, any zone), date32, date64
    c0 = lib.arrow_amd_plugin_calls(b"compare", 1)
    ran_t = 0
    for t in (pa.timestamp("us"), pa.timestamp("ns", tz="UTC"), pa.timestamp("s", tz="Europe/Paris"), pa.duration("ms"),
              pa.time32("s"), pa.time64("ns"), pa.date32(), pa.date64()):
        width = 4 if t in (pa.time32("s"), pa.date32()) else 8
        raw = rng.integers(0, 80_000 if width == 4 else 10**6, n).astype(np.int32 if width == 4 else np.int64)
        if t == pa.date64():
            raw = raw * 86_400_000
        a = pa.array(raw, pa.int32() if width == 4 else pa.int64(), mask=rng.random(n) < 0.06).cast(t)
        b = pa.array(np.roll(raw, 7), pa.int32() if width == 4 else pa.int64(), mask=rng.random(n) < 0.04).cast(t)
        da, db = to_device(a), to_device(b)
        s = a[int(np.flatnonzero(np.asarray(a.is_valid()))[0])]
        for name in cmp_fns:
            fn = lambda x, y: pc.call_function(name, [x, y])
            for dev_out, host_out in ((fn(da, db), fn(a, b)), (fn(da, s), fn(a, s)), (fn(s, db), fn(s, b)),
                                      (fn(da.slice(3, n - 5), db.slice(5, n - 5)), fn(a.slice(3, n - 5), b.slice(5, n - 5)))):
                assert not dev_out.is_cpu, (name, t)
                ho = to_host(dev_out)
                assert ho.equals(host_out) and ho.null_count == host_out.null_count, (name, t)
            ran_t += 4
    assert lib.arrow_amd_plugin_calls(b"compare", 1) - c0 == ran_t
    # a zoned against a zone-less timestamp column is the reference's error, not a comparison of raw integers
    zoned, naive = pa.array([1, 2], pa.timestamp("us", tz="UTC")), pa.array([1, 3], pa.timestamp("us"))
    def message(call):
        try:
            call()
            return None
        except (pa.lib.ArrowInvalid, pa.lib.ArrowTypeError, pa.lib.ArrowNotImplementedError) as e:
            return type(e).__name__ + ": " + str(e)
    want = message(lambda: pc.call_function("less", [zoned, naive]))
    got = message(lambda: pc.call_function("less", [to_device(zoned), to_device(naive)]))
    assert want is not None and got == want, (got, want)
    # an Acero filter + projection over int32 / float32 device columns: (a > 3) & (b < 0.5f) -> a * a + a
    ai = pa.array(rng.integers(-1000, 1000, n).astype(np.int32), mask=rng.random(n) < 0.05)
    bf = pa.array((np.round(rng.standard_normal(n) * 4) / 2).astype(np.float32), mask=rng.random(n) < 0.05)
    t_host, t_dev = pa.table({"a": ai, "b": bf}), pa.table({"a": to_device(ai), "b": to_device(bf)})
    def plan(t):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
            acero.Declaration("filter", acero.FilterNodeOptions((pc.field("a") > pa.scalar(3, pa.int32())) & (pc.field("b") < pa.scalar(0.5, pa.float32())))),
            acero.Declaration("project", acero.ProjectNodeOptions([pc.field("a") * pc.field("a") + pc.field("a"), pc.field("b") * pc.field("b")], ["x", "y"]))])
    got = plan(t_dev).to_table(use_threads=False)
    want = plan(t_host).to_table(use_threads=False)
    assert got.num_rows == want.num_rows
    for name in ("x", "y"):
        g = pa.concat_arrays([c if c.is_cpu else to_host(c) for c in got.column(name).chunks])
        assert g.equals(want.column(name).combine_chunks()), name
    print("NUMERIC_OPS_OK")
''')


AGGREGATE_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    from pyarrow import acero
    rng = np.random.default_rng(23)
    n = SC(3_000_001)
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    smalls = pa.array(rng.integers(-10**6, 10**6, n), mask=rng.random(n) < 0.05)
    d_vals = to_device(vals)
    # scalar aggregates of int64 device columns (sum / count / min_max / min / max): the state is 32 B read back per batch
    red0 = lib.arrow_amd_plugin_calls(b"reduce", 1)
    all_null = pa.array([None] * 1000, pa.int64())
    for h_arr in (vals, smalls, pa.array(rng.integers(-2**62, 2**62, n)), all_null, vals.slice(0, 0)):
        d_arr = to_device(h_arr)
        for opts in (None, pc.ScalarAggregateOptions(skip_nulls=False), pc.ScalarAggregateOptions(min_count=len(h_arr)),
                     pc.ScalarAggregateOptions(min_count=0)):
            for fn in ("sum", "min_max", "min", "max"):
                g, w = pc.call_function(fn, [d_arr], opts), pc.call_function(fn, [h_arr], opts)
                assert g.equals(w) and g.type == w.type, (fn, opts, g, w)
        if len(h_arr) > 200:
            for fn in ("sum", "min_max"):
                assert pc.call_function(fn, [d_arr.slice(13, len(h_arr) - 100)]).equals(pc.call_function(fn, [h_arr.slice(13, len(h_arr) - 100)])), fn
        for mode in ("only_valid", "only_null", "all"):
            assert pc.count(d_arr, mode=mode).equals(pc.count(h_arr, mode=mode)), mode
    # every integer width: narrow columns are widened on the device; sums are int64 / uint64, extrema keep the column's type
    for np_t, lo, hi in ((np.int8, -128, 128), (np.uint8, 0, 256), (np.int16, -2**15, 2**15), (np.uint16, 0, 2**16),
                         (np.int32, -2**31, 2**31), (np.uint32, 0, 2**32)):
        h_arr = pa.array(rng.integers(lo, hi, n // 3).astype(np_t), mask=rng.random(n // 3) < 0.1)
        d_arr = to_device(h_arr)
        for opts in (None, pc.ScalarAggregateOptions(skip_nulls=False), pc.ScalarAggregateOptions(min_count=len(h_arr))):
            for fn in ("sum", "min_max", "min", "max", "mean"):      # (mean: count x max|value| stays below 2^53 here)
                g, w = pc.call_function(fn, [d_arr], opts), pc.call_function(fn, [h_arr], opts)
                assert g.equals(w) and g.type == w.type, (str(h_arr.type), fn, opts, g, w)
        assert pc.min_max(d_arr.slice(7, 1000)).equals(pc.min_max(h_arr.slice(7, 1000)))
        for mode in ("only_valid", "only_null", "all"):
            assert pc.count(d_arr, mode=mode).equals(pc.count(h_arr, mode=mode)), (str(h_arr.type), mode)
    u64 = pa.array(rng.integers(0, 2**64, n // 3, dtype=np.uint64), mask=rng.random(n // 3) < 0.1)
    d_u64 = to_device(u64)
    assert pc.sum(d_u64).equals(pc.sum(u64)) and pc.count(d_u64).equals(pc.count(u64))      # (wraps modulo 2^64 like the reference)
    # the extrema of full-range uint64 (reduced as x + 2^63 read as int64) and the mean of small ones (exact below 2^53)
    assert pc.min_max(d_u64).equals(pc.min_max(u64)) and pc.min(d_u64).equals(pc.min(u64)) and pc.max(d_u64).equals(pc.max(u64))
    u64s = pa.array(rng.integers(0, 2**30, n // 3, dtype=np.uint64), mask=rng.random(n // 3) < 0.1)
    assert pc.mean(to_device(u64s)).equals(pc.mean(u64s)) and pc.mean(to_device(u64s.slice(7, 1001))).equals(pc.mean(u64s.slice(7, 1001)))
    chunks = pa.chunked_array([to_device(vals.slice(0, 1000)), to_device(vals.slice(1000))])       # merge of per-batch states
    assert pc.sum(chunks).equals(pc.sum(vals)) and pc.min_max(chunks).equals(pc.min_max(vals))
    assert lib.arrow_amd_plugin_calls(b"reduce", 1) > red0 + 80
    assert pc.count(pa.array(["a", None])).as_py() == 1 and pc.sum(pa.array([1.5, 2.5])).as_py() == 4.0     # other types: stock
    try:
        pc.sum(pa.chunked_array([vals.slice(0, 10), d_vals]))
        raise SystemExit("expected NotImplemented for a host+device aggregation")
    except pa.lib.ArrowNotImplementedError:
        pass
    # Acero's own ScalarAggregateNode (acero/scalar_aggregate_node.cc) over device batches: table_source -> filter -> aggregate
    m = SC(1_000_003)
    for null_p in (0.0, 0.03):
        mk = (lambda a: pa.array(a, mask=rng.random(m) < null_p)) if null_p else pa.array
        v, w = mk(rng.integers(-2**62, 2**62, m)), mk(rng.integers(-100, 100, m))
        host = pa.table({"v": v, "w": w})
        dev = pa.table({"v": to_device(v), "w": to_device(w)})

        def scalar_plan(table):
            return acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
                acero.Declaration("filter", acero.FilterNodeOptions(pc.field("w") > 10)),
                acero.Declaration("aggregate", acero.AggregateNodeOptions(
                    [("v", "sum", None, "s"), ("v", "min_max", None, "mm"), ("v", "count", None, "c"), ("w", "max", None, "wmax"),
                     ("w", "min", pc.ScalarAggregateOptions(skip_nulls=False), "wmin")]))])
        red0 = lib.arrow_amd_plugin_calls(b"reduce", 1)
        for threads in (False, True):
            assert scalar_plan(dev).to_table(use_threads=threads).equals(scalar_plan(host).to_table(use_threads=threads)), (null_p, threads)
        assert lib.arrow_amd_plugin_calls(b"reduce", 1) >= red0 + 10
    print("AGGREGATE_OK")
''')


ORDER_BY_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))   # sizes shrink for the emulated run
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    from pyarrow import acero
    rng = np.random.default_rng(3)
    n = SC(1_000_003)
    def mk(a, p=0.05): return pa.array(a, mask=rng.random(len(a)) < p)
    k0 = mk(rng.integers(-5, 5, n).astype(np.int32)); k1 = mk(rng.integers(0, 50, n)); 
    fk = np.round(rng.standard_normal(n)*2)/2; fk[rng.random(n)<0.05] = np.nan; k2 = mk(fk)
    v = mk(rng.integers(-2**62, 2**62, n), 0.2); s = pa.array([None if i % 11 == 0 else "s%d" % (i % 1000) for i in range(n)]); b = mk(rng.random(n) < 0.5)
    ts = pa.array(rng.integers(0, 10**6, n), pa.timestamp("us"))
    cols = {"k0": k0, "k1": k1, "k2": k2, "v": v, "s": s, "b": b, "ts": ts}
    host = pa.table(cols); dev = pa.table({k: to_device(a) for k, a in cols.items()})
    def plan(t, node, keys, filt=True):
        seq = [acero.Declaration("table_source", acero.TableSourceNodeOptions(t))]
        if filt: seq.append(acero.Declaration("filter", acero.FilterNodeOptions(pc.field("k1") > 5)))
        seq.append(acero.Declaration(node, acero.OrderByNodeOptions(keys)))
        return acero.Declaration.from_sequence(seq)
    def host_table(t):
        return pa.table({name: pa.chunked_array([c if c.is_cpu else to_host(c) for c in t.column(name).chunks], t.schema.field(name).type) for name in t.schema.names})
    def same(a, b):
        assert a.schema == b.schema and a.num_rows == b.num_rows, (a.schema, b.schema, a.num_rows, b.num_rows)
        for name in a.schema.names:
            x, y = a.column(name).combine_chunks(), b.column(name).combine_chunks()
            if pa.types.is_floating(x.type):
                assert np.array_equal(np.asarray(x.is_null()), np.asarray(y.is_null()))
                x, y = (pc.fill_null(z, 0.0).to_numpy(zero_copy_only=False).view(np.uint64) for z in (x, y))
                assert np.array_equal(x, y), name
            else:
                assert x.equals(y), name
    light = os.environ.get("ARROW_AMD_TEST_LIGHT") == "1"      # (every sort launch costs seconds under the emulator)
    for keys in ([("k0", "ascending"), ("k1", "descending")], [("k2", "descending", "at_start"), ("k0", "ascending", "at_end"), ("ts", "ascending")], [("k1", "ascending")])[: 2 if light else 3]:
        for filt in (((False,) if len(keys) == 2 else (True,)) if light else (True, False)):      # (unfiltered: chunks are consecutive slices, re-joined without a copy)
            want = plan(host, "order_by", keys, filt).to_table(use_threads=False)
            for th in ((True,) if light else (False, True)):
                got = plan(dev, "order_by_rocm", keys, filt).to_table(use_threads=th)
                assert not got.column("v").chunk(0).is_cpu
                same(host_table(got), want)
            goth = plan(host, "order_by_rocm", keys, filt).to_table(use_threads=True)
            assert goth.column("v").chunk(0).is_cpu
            same(goth, want)
    print("gpu", lib.arrow_amd_plugin_calls(b"order_by",1))
    # empty input
    got = acero.Declaration.from_sequence([acero.Declaration("table_source", acero.TableSourceNodeOptions(dev)),
        acero.Declaration("filter", acero.FilterNodeOptions(pc.field("k1") > 1000)),
        acero.Declaration("order_by_rocm", acero.OrderByNodeOptions([("k0","ascending")]))]).to_table()
    assert got.num_rows == 0 and got.schema == host.schema
    # (round 5) utf8 keys: the string as (8-byte big-endian chunks, length) keys of the same chain; equal to the stock order_by
    for keys in ([("s", "ascending")], [("s", "descending"), ("k0", "ascending")]):
        want = plan(host, "order_by", keys).to_table(use_threads=False)
        same(host_table(plan(dev, "order_by_rocm", keys).to_table(use_threads=False)), want)
    # array_sort_indices / sort_indices of device-resident utf8 / binary ARRAYS: common prefixes longer than one chunk, strings
    # that differ only in trailing NUL bytes or only in length, empty vs null, bytes >= 0x80 (unsigned order), both orders and
    # null placements; a string beyond 256 bytes is refused by name
    words = ["", "a", "ab", "ab\x00", "ab\x00\x00", "abcdefgh", "abcdefghi", "abcdefgh\x00", "abcdefghijklmnopq", "abcdefghijklmnopr",
             "zz", "\u00e9t\u00e9", "\u4e2d\u6587", "Z", "abc", "abd", "ab" * 40, "ab" * 40 + "c"]
    pick = rng.integers(0, len(words), 4000)
    for typ in (pa.utf8(), pa.binary()):
        vals = [words[i] if typ == pa.utf8() else words[i].encode("utf8") for i in pick]
        harr = pa.array(vals, typ, mask=rng.random(len(vals)) < 0.1).slice(3)
        darr = to_device(harr)
        for order in ("ascending", "descending"):
            for placement in ("at_end", "at_start"):
                w = pc.array_sort_indices(harr, order=order, null_placement=placement)
                g = to_host(pc.array_sort_indices(darr, order=order, null_placement=placement))
                assert g.equals(w), (str(typ), order, placement, g.slice(0, 8), w.slice(0, 8))
        assert to_host(pc.sort_indices(darr)).equals(pc.sort_indices(harr))
    assert to_host(pc.array_sort_indices(to_device(pa.array([], pa.utf8())))).equals(pc.array_sort_indices(pa.array([], pa.utf8())))
    assert to_host(pc.array_sort_indices(to_device(pa.array([None, None], pa.utf8())))).equals(pc.array_sort_indices(pa.array([None, None], pa.utf8())))
    try:
        pc.array_sort_indices(to_device(pa.array(["x" * 300, "y"])))
        raise SystemExit("a 300-byte sort key was accepted")
    except pa.ArrowNotImplementedError as ex:
        assert "256 bytes" in str(ex), ex
    assert pc.array_sort_indices(pa.array(["x" * 300, "y"])).to_pylist() == [0, 1]          # host arrays: the reference's kernel
    print("ORDER_BY_OK")
''')


PARQUET_SCRIPT = textwrap.dedent(r'''
    import ctypes, faulthandler, os, sys, tempfile
    import numpy as np
    import pyarrow as pa, pyarrow.parquet as pq
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    def read_column(path, rg, col):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        rc = lib.arrow_amd_parquet_read_column(path.encode(), rg, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))

    rng = np.random.default_rng(17)
    n = SC(1_000_003)
    m = lambda p: (rng.random(n) < p) if p else None
    for null_p in (0.0, 0.12):
        t = pa.table({"few": pa.array(rng.integers(-50, 50, n), mask=m(null_p)),
                      "wide": pa.array(rng.integers(-2**62, 2**62, n), mask=m(null_p)),
                      "grow": pa.array(np.arange(n) // 3, mask=m(null_p)),
                      "i32": pa.array(np.repeat(rng.integers(0, 9, n // 50 + 1), 50)[:n].astype(np.int32), mask=m(null_p)),
                      "f64": pa.array(np.round(rng.standard_normal(n), 2), mask=m(null_p)),
                      "ts": pa.array(rng.integers(0, 2**50, n), pa.timestamp("us", tz="UTC"), mask=m(null_p)),     # logical types with the
                      "day": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m(null_p)),   # physical layout: labelled
                      "tod": pa.array(rng.integers(0, 86_400_000, n).astype(np.int32), pa.time32("ms"), mask=m(null_p)),
                      "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m(null_p)),
                      "flag": pa.array(rng.random(n) < 0.3, type=pa.bool_(), mask=m(null_p)),
                      "flag_runs": pa.array(np.repeat(rng.random(n // 40 + 1) < 0.5, 40)[:n], type=pa.bool_(), mask=m(null_p)),
                      "str": pa.array(np.array(["", "a", "bb", "gfx950", "MI355X", "ünïcödé", "x" * 40], dtype=object)[rng.integers(0, 7, n)],
                                      type=pa.string(), mask=m(null_p)),
                      "str_wide": pa.array(np.array([("w%d" % i) * (i % 4) for i in range(n)], dtype=object)[rng.integers(0, n, n)],
                                           type=pa.string(), mask=m(null_p)),
                      "bin": pa.array([bytes([i % 251]) * (i % 6) for i in range(n)], type=pa.binary(), mask=m(null_p))})
        for variant in (dict(compression="snappy"), dict(compression="zstd", data_page_version="2.0", data_page_size=8192),
                        dict(compression="none", use_dictionary=False),
                        dict(compression="snappy", dictionary_pagesize_limit=16384, data_page_size=8192)):
            path = os.path.join(tempfile.mkdtemp(), "t.parquet")
            pq.write_table(t, path, row_group_size=n // 2 + 11, **variant)
            pf = pq.ParquetFile(path)
            for rg in range(pf.metadata.num_row_groups):
                ref = pf.read_row_group(rg)
                for ci, name in enumerate(t.schema.names):
                    d = read_column(path, rg, ci)
                    assert not d.is_cpu, name
                    h = to_host(d)
                    w = ref.column(name).combine_chunks()
                    assert h.equals(w) and h.null_count == w.null_count, (variant, null_p, rg, name, h.slice(0, 5), w.slice(0, 5))
    assert lib.arrow_amd_plugin_calls(b"parquet", 1) > 0
    # Snappy chunks are read raw and their PLAIN value pages decompressed on the device (V2 pages: the values behind the
    # levels; V1 pages of required columns: the whole body); dictionary pages / V1 optional pages / other encodings are
    # decompressed on the host behind the same raw reader.  Same arrays with the route switched off.
    lib.arrow_amd_plugin_parquet_device_snappy_pages.restype = ctypes.c_int64
    req = pa.table({"a": pa.array(np.cumsum(rng.integers(-3, 4, n))), "b": pa.array(rng.integers(0, 50, n).astype(np.int32)),
                    "c": pa.array(np.round(rng.standard_normal(n), 1)), "d": pa.array(rng.integers(0, 7, n), mask=rng.random(n) < 0.1)})
    req = req.cast(pa.schema([pa.field("a", pa.int64(), nullable=False), pa.field("b", pa.int32(), nullable=False),
                              pa.field("c", pa.float64(), nullable=False), pa.field("d", pa.int64())]))
    for variant in (dict(data_page_version="1.0", use_dictionary=False, data_page_size=16384),
                    dict(data_page_version="2.0", use_dictionary=False, data_page_size=16384),
                    dict(data_page_version="2.0", use_dictionary=["b", "d"]),
                    dict(data_page_version="1.0", use_dictionary=["a"], dictionary_pagesize_limit=4096, data_page_size=8192)):
        path = os.path.join(tempfile.mkdtemp(), "r.parquet")
        pq.write_table(req, path, row_group_size=n // 2 + 11, compression="snappy", **variant)
        pf = pq.ParquetFile(path)
        # (pinned: the chunk's bytes in page-locked / pageable host memory; threads: the chunk read in that many parts)
        for on, pinned, threads in ((1, 1, 1), (1, 1, 3), (1, 0, 2), (0, 1, 1)):
            lib.arrow_amd_plugin_set_parquet_device_snappy(on)
            lib.arrow_amd_plugin_set_parquet_pinned_staging(pinned)
            lib.arrow_amd_plugin_set_parquet_read_threads(threads, ctypes.c_int64(4096))
            before = lib.arrow_amd_plugin_parquet_device_snappy_pages()
            for rg in range(pf.metadata.num_row_groups):
                ref = pf.read_row_group(rg)
                for ci, name in enumerate(req.schema.names):
                    h = to_host(read_column(path, rg, ci))
                    w = ref.column(name).combine_chunks()
                    assert h.equals(w) and h.null_count == w.null_count, (variant, on, pinned, threads, rg, name)
            used = lib.arrow_amd_plugin_parquet_device_snappy_pages() - before
            assert (used > 0) if on else (used == 0), (variant, on, used)
    lib.arrow_amd_plugin_set_parquet_device_snappy(1)
    lib.arrow_amd_plugin_set_parquet_pinned_staging(1)
    lib.arrow_amd_plugin_set_parquet_read_threads(4, ctypes.c_int64(1 << 23))
    # GZIP chunks take the same raw route (round 6: arx_gzip_decompress_pages — RFC 1952 / 1951 on the device); dictionary pages
    # and V1 optional pages are inflated on the host behind the same reader.  Same arrays with the route switched off.
    lib.arrow_amd_plugin_parquet_device_gzip_pages.restype = ctypes.c_int64
    for variant in (dict(data_page_version="2.0", use_dictionary=False, data_page_size=16384),
                    dict(data_page_version="1.0", use_dictionary=["b", "d"], data_page_size=16384)):
        path = os.path.join(tempfile.mkdtemp(), "g.parquet")
        req_g = req.slice(0, n // 4)                 # (the host's deflate is the slow part of this check)
        pq.write_table(req_g, path, row_group_size=n // 8 + 11, compression="gzip", **variant)
        pf = pq.ParquetFile(path)
        for on in (1, 0):
            lib.arrow_amd_plugin_set_parquet_device_gzip(on)
            before = lib.arrow_amd_plugin_parquet_device_gzip_pages()
            for rg in range(pf.metadata.num_row_groups):
                ref = pf.read_row_group(rg)
                for ci, name in enumerate(req.schema.names):
                    h = to_host(read_column(path, rg, ci))
                    w = ref.column(name).combine_chunks()
                    assert h.equals(w) and h.null_count == w.null_count, ("gzip", variant, on, rg, name)
            used = lib.arrow_amd_plugin_parquet_device_gzip_pages() - before
            assert (used > 0) if on else (used == 0), ("gzip", variant, on, used)
    lib.arrow_amd_plugin_set_parquet_device_gzip(1)
    # a garbled GZIP page: an IOError that names zlib's inflate, from either side
    path = os.path.join(tempfile.mkdtemp(), "badgz.parquet")
    pq.write_table(req.select(["a"]), path, compression="gzip", use_dictionary=False, data_page_version="2.0")
    raw = bytearray(open(path, "rb").read())
    off = pq.ParquetFile(path).metadata.row_group(0).column(0).data_page_offset
    from arrow_amd.parquet import read_page_header
    hdr, body = read_page_header(bytes(raw), off)
    for k in range(body + hdr[3] // 3, body + hdr[3] // 3 + 64):
        raw[k] ^= 0x5A
    open(path, "wb").write(bytes(raw))
    try:
        want, ref_error = pq.ParquetFile(path).read_row_group(0).column("a").combine_chunks(), None
    except Exception as e:
        want, ref_error = None, str(e)
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    rc = lib.arrow_amd_parquet_read_column(path.encode(), 0, 0, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    if ref_error is not None:
        assert rc != 0 and b"zlib inflate failed" in lib.arrow_amd_plugin_last_error(), (ref_error, lib.arrow_amd_plugin_last_error())
    else:
        assert rc == 0 and to_host(pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))).equals(want)
    lib.arrow_amd_plugin_parquet_copied_pages.restype = ctypes.c_int64
    assert lib.arrow_amd_plugin_parquet_copied_pages() == 0    # every device-route page was used where the chunk read put it
    # a corrupt Snappy page is reported with the reference's text, whichever side decompresses it
    path = os.path.join(tempfile.mkdtemp(), "bad.parquet")
    pq.write_table(req.select(["a"]), path, compression="snappy", use_dictionary=False, data_page_version="2.0")
    raw = bytearray(open(path, "rb").read())
    off = pq.ParquetFile(path).metadata.row_group(0).column(0).data_page_offset
    from arrow_amd.parquet import read_page_header
    hdr, body = read_page_header(bytes(raw), off)      # the first data page: garble the middle of its compressed body
    for k in range(body + hdr[3] // 3, body + 2 * hdr[3] // 3):
        raw[k] ^= 0x5A
    open(path, "wb").write(bytes(raw))
    try:            # (Snappy has no checksum: garbage may also decode to other bytes — then both sides must agree on them)
        want, ref_error = pq.ParquetFile(path).read_row_group(0).column("a").combine_chunks(), None
    except Exception as e:
        want, ref_error = None, str(e)
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    rc = lib.arrow_amd_parquet_read_column(path.encode(), 0, 0, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    if ref_error is not None:
        assert "orrupt snappy" in ref_error, ref_error
        assert rc != 0 and b"Corrupt snappy compressed data" in lib.arrow_amd_plugin_last_error(), lib.arrow_amd_plugin_last_error()
    else:
        assert rc == 0 and to_host(pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))).equals(want)
    # several chunks of a row group at once (a pool of worker threads, each with its own stream and buffers): the same
    # arrays as one call per column, twice (the second call finds the workers' buffers warm); an error in one column
    # fails the call
    path = os.path.join(tempfile.mkdtemp(), "m.parquet")
    pq.write_table(req, path, row_group_size=n // 2 + 11, compression="snappy", data_page_version="2.0", use_dictionary=["b", "d"])
    pf = pq.ParquetFile(path)
    ncol = len(req.schema.names)
    IntArr = ctypes.c_int * (2 * ncol)
    order = list(range(ncol)) + list(reversed(range(ncol)))          # (every column twice: more tasks than workers)
    lib.arrow_amd_parquet_read_columns.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for _ in range(2):
        for rg in range(pf.metadata.num_row_groups):
            ref = pf.read_row_group(rg)
            c_devs, c_schemas = ctypes.create_string_buffer(128 * len(order)), ctypes.create_string_buffer(72 * len(order))
            rc = lib.arrow_amd_parquet_read_columns(path.encode(), rg, IntArr(*order), len(order), ctypes.addressof(c_devs), ctypes.addressof(c_schemas))
            assert rc == 0, lib.arrow_amd_plugin_last_error()
            for i, ci in enumerate(order):
                d = pa.Array._import_from_c_device(ctypes.addressof(c_devs) + 128 * i, ctypes.addressof(c_schemas) + 72 * i)
                w = ref.column(req.schema.names[ci]).combine_chunks()
                h = to_host(d)
                assert h.equals(w) and h.null_count == w.null_count, (rg, ci)
    c_devs, c_schemas = ctypes.create_string_buffer(128 * 3), ctypes.create_string_buffer(72 * 3)
    assert lib.arrow_amd_parquet_read_columns(path.encode(), 0, (ctypes.c_int * 3)(0, 99, 1), 3, ctypes.addressof(c_devs), ctypes.addressof(c_schemas)) != 0
    assert b"no row group" in lib.arrow_amd_plugin_last_error() or b"column" in lib.arrow_amd_plugin_last_error()
    # a struct column is refused, not mis-decoded (lists of primitives are read: parquet_list_columns_through_the_plugin)
    path = os.path.join(tempfile.mkdtemp(), "l.parquet")
    pq.write_table(pa.table({"l": pa.array([{"a": 1}, None])}), path)
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    assert lib.arrow_amd_parquet_read_column(path.encode(), 0, 0, ctypes.addressof(c_dev), ctypes.addressof(c_schema)) != 0
    assert b"structs are not on the device path" in lib.arrow_amd_plugin_last_error(), lib.arrow_amd_plugin_last_error()
    print("PARQUET_OK")
''')


PARQUET_ENCODINGS_SCRIPT = textwrap.dedent(r'''
    import ctypes, faulthandler, os, sys, tempfile
    import numpy as np
    import pyarrow as pa, pyarrow.parquet as pq
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    def read_column(path, rg, col):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        rc = lib.arrow_amd_parquet_read_column(path.encode(), rg, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))

    rng = np.random.default_rng(17)
    n = SC(1_000_003)
    m = lambda p: (rng.random(n) < p) if p else None
    for null_p in (0.0, 0.12):
        # DELTA_BINARY_PACKED integer columns (sorted ids, a random walk, wrap-around deltas, constants)
        dt = pa.table({"sorted64": pa.array(np.sort(rng.integers(0, 2**40, n)), mask=m(null_p)),
                       "walk32": pa.array(np.cumsum(rng.integers(-50, 60, n)).astype(np.int32), mask=m(null_p)),
                       "full64": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=m(null_p)),
                       "const32": pa.array(np.full(n, 7, dtype=np.int32), mask=m(null_p))})
        for variant in (dict(compression="snappy"), dict(compression="none", data_page_version="2.0", data_page_size=4096)):
            path = os.path.join(tempfile.mkdtemp(), "d.parquet")
            pq.write_table(dt, path, row_group_size=n // 2 + 11, use_dictionary=False,
                           column_encoding={name: "DELTA_BINARY_PACKED" for name in dt.schema.names}, **variant)
            pf = pq.ParquetFile(path)
            for rg in range(pf.metadata.num_row_groups):
                ref = pf.read_row_group(rg)
                for ci, name in enumerate(dt.schema.names):
                    assert "DELTA_BINARY_PACKED" in pf.metadata.row_group(rg).column(ci).encodings
                    h = to_host(read_column(path, rg, ci))
                    w = ref.column(name).combine_chunks()
                    assert h.equals(w) and h.null_count == w.null_count, ("delta", variant, null_p, rg, name)
        # DELTA_LENGTH_BYTE_ARRAY strings
        lt = pa.table({"s": pa.array(np.array([("w%d" % i) * (i % 5) for i in range(n)], dtype=object)[rng.integers(0, n, n)],
                                     type=pa.string(), mask=m(null_p)),
                       "b": pa.array([bytes([i % 251]) * (i % 9) for i in range(n)], type=pa.binary(), mask=m(null_p))})
        path = os.path.join(tempfile.mkdtemp(), "dl.parquet")
        pq.write_table(lt, path, row_group_size=n // 2 + 11, use_dictionary=False, data_page_size=32768,
                       column_encoding={name: "DELTA_LENGTH_BYTE_ARRAY" for name in lt.schema.names})
        pf = pq.ParquetFile(path)
        for rg in range(pf.metadata.num_row_groups):
            ref = pf.read_row_group(rg)
            for ci, name in enumerate(lt.schema.names):
                assert "DELTA_LENGTH_BYTE_ARRAY" in pf.metadata.row_group(rg).column(ci).encodings
                h = to_host(read_column(path, rg, ci))
                w = ref.column(name).combine_chunks()
                assert h.equals(w) and h.null_count == w.null_count, ("delta_length", null_p, rg, name)
        # DELTA_BYTE_ARRAY (DeltaByteArrayDecoderImpl): sorted keys with long shared prefixes, repeats and shrinking values,
        # values longer than the kernel's 8 KB LDS window; many small pages (every page starts from the empty string)
        long_ = bytes(rng.integers(97, 123, 20000, dtype=np.uint8))
        bt = pa.table({"sorted": pa.array(sorted("key/%08d/%s" % (int(k), "x" * int(k % 7)) for k in rng.integers(0, 10 * n, n)),
                                          type=pa.string(), mask=m(null_p)),
                       "mixed": pa.array([[b"", b"a", b"ab", b"abc" * 11, b"abc" * 11 + b"d"][int(i)] for i in rng.integers(0, 5, n)],
                                         type=pa.binary(), mask=m(null_p)),
                       "long": pa.array([long_[: int(k)] + bytes([65 + int(k) % 26])
                                         for k in rng.choice([10, 8191, 8192, 8193, 19999], n, p=[0.96, 0.01, 0.01, 0.01, 0.01])], type=pa.binary())})
        for variant in (dict(compression="snappy", data_page_size=16384), dict(compression="none", data_page_version="2.0", data_page_size=4096)):
            path = os.path.join(tempfile.mkdtemp(), "dba.parquet")
            pq.write_table(bt, path, row_group_size=n // 2 + 11, use_dictionary=False,
                           column_encoding={name: "DELTA_BYTE_ARRAY" for name in bt.schema.names}, **variant)
            pf = pq.ParquetFile(path)
            for rg in range(pf.metadata.num_row_groups):
                ref = pf.read_row_group(rg)
                for ci, name in enumerate(bt.schema.names):
                    assert "DELTA_BYTE_ARRAY" in pf.metadata.row_group(rg).column(ci).encodings
                    h = to_host(read_column(path, rg, ci))
                    w = ref.column(name).combine_chunks()
                    assert h.equals(w) and h.null_count == w.null_count, ("delta_byte_array", variant, null_p, rg, name)
        # BYTE_STREAM_SPLIT floating-point and integer columns
        st = pa.table({"f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m(null_p)),
                       "f64": pa.array(rng.standard_normal(n) * 1e100, mask=m(null_p)),
                       "i64": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=m(null_p))})
        path = os.path.join(tempfile.mkdtemp(), "s.parquet")
        pq.write_table(st, path, row_group_size=n // 2 + 11, use_dictionary=False, data_page_size=65536,
                       column_encoding={name: "BYTE_STREAM_SPLIT" for name in st.schema.names})
        pf = pq.ParquetFile(path)
        for rg in range(pf.metadata.num_row_groups):
            ref = pf.read_row_group(rg)
            for ci, name in enumerate(st.schema.names):
                assert "BYTE_STREAM_SPLIT" in pf.metadata.row_group(rg).column(ci).encodings
                h = to_host(read_column(path, rg, ci))
                w = ref.column(name).combine_chunks()
                assert h.equals(w) and h.null_count == w.null_count, ("byte_stream_split", null_p, rg, name)
    assert lib.arrow_amd_plugin_calls(b"parquet", 1) > 0
    print("PARQUET_ENCODINGS_OK")
''')


TABLE_SOURCE_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(77)
    n = SC(3_000_000)
    t = pa.table({"x": pa.array(rng.random(n)),
                  "k": pa.array(rng.integers(-5000, 5000, n).astype(np.int32)),
                  "v": pa.array(rng.integers(-2**60, 2**60, n), mask=rng.random(n) < 0.1)})
    pred = pc.field("x") > 0.75
    def plan(source, tab, tail):
        return acero.Declaration.from_sequence([acero.Declaration(source, acero.TableSourceNodeOptions(tab))] + tail)
    def filter_project():
        return [acero.Declaration("filter", acero.FilterNodeOptions(pred)),
                acero.Declaration("project", acero.ProjectNodeOptions([pc.field("k"), pc.add(pc.field("v"), pc.field("v"))], ["k", "w"]))]
    # ---- the reference: its own source, nodes and kernels, before anything is registered
    want_rows = plan("table_source", t, filter_project()).to_table(use_threads=False)
    want_groups = t.filter(pc.greater(t.column("x"), 0.75)).group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k")
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    for f in ("arrow_amd_plugin_aggregate_direct_batches", "arrow_amd_plugin_aggregate_flushes"):
        getattr(lib, f).restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.arrow_amd_plugin_set_table_source_rows.argtypes = [ctypes.c_int64]
    lib.arrow_amd_plugin_set_aggregate_direct_rows.argtypes = [ctypes.c_int64]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(arr):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        arr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), arr.type)
    def host_table(tab):
        return pa.table({name: pa.chunked_array([to_host(c) for c in tab.column(name).chunks], tab.schema.field(name).type)
                         for name in tab.schema.names})

    # two chunks of different sizes, every column in HBM
    cut = n // 3 + 5
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names) for c in (t.slice(0, cut), t.slice(cut))])
    # ---- table_source_rocm -> filter -> project: the STOCK FilterNode / ProjectNode run once per chunk
    f0 = lib.arrow_amd_plugin_calls(b"array_filter", 1)
    got = plan("table_source_rocm", td, filter_project()).to_table(use_threads=False)
    launches = lib.arrow_amd_plugin_calls(b"array_filter", 1) - f0
    assert launches == 2 * 3, ("one array_filter per column and chunk, not per 32Ki rows", launches)
    got = host_table(got)
    assert got.schema.equals(want_rows.schema), (got.schema, want_rows.schema)
    assert got.equals(want_rows), "rows, values and ORDER of the reference plan"       # implicit ordering + batch indices
    # batches of a few thousand rows (the knob that replaces the options' default): more, smaller batches, same rows
    lib.arrow_amd_plugin_set_table_source_rows(SC(400_000) + 3)
    f0 = lib.arrow_amd_plugin_calls(b"array_filter", 1)
    got = host_table(plan("table_source_rocm", td, filter_project()).to_table(use_threads=False))
    assert got.equals(want_rows)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) - f0 > 2 * 3
    lib.arrow_amd_plugin_set_table_source_rows(1 << 27)
    # (round 6: the stock table_source delivers whole chunks of a device table by default; the opt-out brings the reference
    #  SourceNode's morsels back for this part)
    assert lib.arrow_amd_override_acero_factories(-1) == 0, lib.arrow_amd_plugin_last_error()
    # ---- coalesce_rocm behind the STOCK source: its 32Ki-row batches are joined again before the filter sees them
    # (consecutive slices of one device array: no copy); the rows, values and order of the reference plan
    lib.arrow_amd_plugin_coalesced_batches.restype = ctypes.c_int64
    lib.arrow_amd_plugin_set_coalesce_rows.argtypes = [ctypes.c_int64]
    any_options = acero.FilterNodeOptions(pc.scalar(True))      # (the node takes no options; pyarrow needs an object)
    def coalesced(rows):
        lib.arrow_amd_plugin_set_coalesce_rows(rows)
        c0, f0 = lib.arrow_amd_plugin_coalesced_batches(), lib.arrow_amd_plugin_calls(b"array_filter", 1)
        got = host_table(plan("table_source", td, [acero.Declaration("coalesce_rocm", any_options)] + filter_project()).to_table(use_threads=False))
        return got, lib.arrow_amd_plugin_coalesced_batches() - c0, lib.arrow_amd_plugin_calls(b"array_filter", 1) - f0
    stock_batches = -(-cut // 32768) + -(-(n - cut) // 32768)
    got, joined, launches = coalesced(1 << 26)
    assert got.equals(want_rows), "coalesce_rocm must not change rows, values or order"
    assert joined == stock_batches and launches == 3, (joined, stock_batches, launches)     # one batch, one filter per column
    got, joined, launches = coalesced(SC(500_000))
    assert got.equals(want_rows) and 3 < launches <= 3 * stock_batches, (launches, stock_batches)    # several batches, fewer than the source's at full size
    lib.arrow_amd_plugin_set_coalesce_rows(1 << 26)
    # behind a FILTER the batches are separate buffers: the copying path of the concatenation — fixed-width values with
    # nulls, bit-packed booleans, utf8 offsets rebased and bytes appended
    m = SC(400_000)
    t2 = pa.table({"x": pa.array(rng.random(m)),
                   "i": pa.array(rng.integers(-2**62, 2**62, m), mask=rng.random(m) < 0.1),
                   "b": pa.array(rng.random(m) < 0.5, mask=rng.random(m) < 0.2),
                   "s": pa.array(["k%d" % (i % 1013) if i % 5 else None for i in range(m)])})
    want2 = t2.filter(pc.greater(t2.column("x"), 0.5))
    td2 = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(t2.column(j).chunk(0)) for j in range(t2.num_columns)], names=t2.schema.names)])
    lib.arrow_amd_plugin_set_coalesce_rows(1 << 26)
    c0 = lib.arrow_amd_plugin_coalesced_batches()
    got2 = plan("table_source", td2, [acero.Declaration("filter", acero.FilterNodeOptions(pc.field("x") > 0.5)),
                                      acero.Declaration("coalesce_rocm", any_options)]).to_table(use_threads=False)
    nb2 = -(-m // 32768)
    assert got2.num_rows == want2.num_rows and all(c.num_chunks == 1 for c in got2.columns), "one batch out"
    assert lib.arrow_amd_plugin_coalesced_batches() - c0 == (nb2 if nb2 > 1 else 0)      # (a single batch is handed on as it came)
    got2 = host_table(got2)
    for name in t2.schema.names:
        assert got2.column(name).combine_chunks().equals(want2.column(name).combine_chunks()), name
    # host batches pass through untouched (nothing is moved to the device behind the caller's back)
    c0 = lib.arrow_amd_plugin_coalesced_batches()
    got = plan("table_source", t, [acero.Declaration("coalesce_rocm", any_options)] + filter_project()).to_table(use_threads=False)
    assert got.equals(want_rows) and lib.arrow_amd_plugin_coalesced_batches() == c0
    assert lib.arrow_amd_override_acero_factories(0) == 0
    # ---- ... -> aggregate_rocm: large batches are consumed where they lie (no staging copy)
    lib.arrow_amd_plugin_set_aggregate_direct_rows(SC(100_000))
    agg = [acero.Declaration("filter", acero.FilterNodeOptions(pred)),
           acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))]
    d0, fl0 = lib.arrow_amd_plugin_aggregate_direct_batches(), lib.arrow_amd_plugin_aggregate_flushes()
    got = plan("table_source_rocm", td, agg).to_table(use_threads=False).sort_by("k")
    assert lib.arrow_amd_plugin_aggregate_direct_batches() - d0 == 2 and lib.arrow_amd_plugin_aggregate_flushes() == fl0
    assert got.column("k").equals(want_groups.column("k")) and got.column("v_sum").equals(want_groups.column("v_sum")), (got.slice(0, 5), want_groups.slice(0, 5))
    # small and large batches mixed: the small ones are staged, the large ones are not; one result
    lib.arrow_amd_plugin_set_aggregate_direct_rows(cut * 3 // 8)  # (a quarter of the rows pass: the first chunk leaves ~cut / 4 rows, the second ~cut / 2)
    d0 = lib.arrow_amd_plugin_aggregate_direct_batches()
    got = plan("table_source_rocm", td, agg).to_table(use_threads=False).sort_by("k")
    assert lib.arrow_amd_plugin_aggregate_direct_batches() - d0 == 1
    assert got.column("k").equals(want_groups.column("k")) and got.column("v_sum").equals(want_groups.column("v_sum"))
    # (round 5) large batches that are CONSECUTIVE SLICES of one device column are one span: the source cuts the unfiltered
    # table into several batches, aggregate_rocm runs ONE partitioned pass over all of them (hash_sum GPU calls: one per chunk
    # of the table, not one per batch)
    lib.arrow_amd_plugin_set_aggregate_direct_rows(SC(20_000))
    lib.arrow_amd_plugin_set_table_source_rows(SC(50_000))
    plain = [acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))]
    want_plain = t.group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k")
    d0, h0 = lib.arrow_amd_plugin_aggregate_direct_batches(), lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    got = plan("table_source_rocm", td, plain).to_table(use_threads=False).sort_by("k")
    assert got.column("k").equals(want_plain.column("k")) and got.column("v_sum").equals(want_plain.column("v_sum"))
    assert lib.arrow_amd_plugin_aggregate_direct_batches() - d0 >= 4, lib.arrow_amd_plugin_aggregate_direct_batches() - d0
    # (+ 1: a chunk's last, short batch is staged with the small ones and consumed at the end)
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - h0 <= td.column("k").num_chunks + 1 < lib.arrow_amd_plugin_aggregate_direct_batches() - d0, (
        lib.arrow_amd_plugin_calls(b"hash_sum", 1) - h0, td.column("k").num_chunks, lib.arrow_amd_plugin_aggregate_direct_batches() - d0)
    # (round 6) a plan that is ONE run of device slices without nulls, sums and counts only: the range-partitioned state
    # (arx_groupby_range_*) — sampled key range, no table; what it declines (a wide key range here) goes through the table.
    # Same groups, sums, counts and min_count validity as the reference either way.
    for f in ("arrow_amd_plugin_aggregate_range_plans", "arrow_amd_plugin_aggregate_range_declined"):
        getattr(lib, f).restype = ctypes.c_int64
    lib.arrow_amd_plugin_set_aggregate_range_min_rows.argtypes = [ctypes.c_int64]
    lib.arx_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    for name, value in ((b"groupby_lines_wgs", 2), (b"groupby_lines_unit_rows", 4096)):      # (sized for this test's rows)
        assert lib.arx_set_option(name, value) == 0
    m3 = SC(120_000)
    lib.arrow_amd_plugin_set_aggregate_range_min_rows(m3 // 2)
    opts3 = pc.ScalarAggregateOptions(min_count=9)
    agg3 = [acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions(
        [("v", "hash_sum", opts3, "v_sum"), ("v", "hash_count", None, "v_count"), ("v", "hash_sum", None, "v_all")], keys=["k"]))]
    for lo, hi, through_range in ((-5000, 5000, True), (-2**31, 2**31 - 1, False)):
        t3 = pa.table({"k": pa.array(rng.integers(lo, hi, m3).astype(np.int32)), "v": pa.array(rng.integers(-2**60, 2**60, m3))})
        want3 = t3.group_by("k", use_threads=False).aggregate([("v", "sum", opts3), ("v", "count"), ("v", "sum")]).sort_by("k")
        td3 = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(t3.column(j).chunk(0)) for j in range(2)], names=t3.schema.names)])
        r0, x0 = lib.arrow_amd_plugin_aggregate_range_plans(), lib.arrow_amd_plugin_aggregate_range_declined()
        got3 = plan("table_source_rocm", td3, agg3).to_table(use_threads=False).sort_by("k")
        assert lib.arrow_amd_plugin_aggregate_range_plans() - r0 == (1 if through_range else 0), (lo, hi)
        assert got3.column("k").equals(want3.column("k")), (lo, hi)
        # (want3's aggregate columns: v_sum, v_count, v_sum again — in that order, before or behind the key column)
        w_sum, w_count, w_all = [want3.column(j) for j, name in enumerate(want3.schema.names) if name != "k"]
        assert got3.column("v_sum").equals(w_sum), (lo, hi, got3.column("v_sum").null_count, w_sum.null_count)
        assert got3.column("v_count").equals(w_count), (lo, hi)
        assert got3.column("v_all").equals(w_all), (lo, hi)
        assert w_sum.null_count > 0 and w_all.null_count == 0      # (min_count = 9 bites: ~12 rows a group at most)
    # a hot key makes the scatter give up: declined, and the table takes the rows
    t3 = pa.table({"k": pa.array(np.where(rng.random(m3) < 0.9, 17, rng.integers(0, 9000, m3)).astype(np.int32)), "v": pa.array(rng.integers(-2**60, 2**60, m3))})
    want3 = t3.group_by("k", use_threads=False).aggregate([("v", "sum")]).sort_by("k")
    td3 = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(t3.column(j).chunk(0)) for j in range(2)], names=t3.schema.names)])
    r0, x0 = lib.arrow_amd_plugin_aggregate_range_plans(), lib.arrow_amd_plugin_aggregate_range_declined()
    got3 = plan("table_source_rocm", td3, plain).to_table(use_threads=False).sort_by("k")
    assert got3.column("k").equals(want3.column("k")) and got3.column("v_sum").equals(want3.column("v_sum"))
    assert lib.arrow_amd_plugin_aggregate_range_plans() - r0 + lib.arrow_amd_plugin_aggregate_range_declined() - x0 == 1
    lib.arrow_amd_plugin_set_aggregate_range_min_rows(1 << 22)
    for name, value in ((b"groupby_lines_wgs", 0), (b"groupby_lines_unit_rows", 1 << 21)):
        assert lib.arx_set_option(name, value) == 0
    lib.arrow_amd_plugin_set_table_source_rows(1 << 27)
    lib.arrow_amd_plugin_set_aggregate_direct_rows(1 << 22)
    # the result may stay in HBM when the rows came from there (off by default: GroupByNode's result is host memory)
    lib.arrow_amd_plugin_set_aggregate_device_output(1)
    got_d = plan("table_source_rocm", td, agg).to_table(use_threads=False)
    lib.arrow_amd_plugin_set_aggregate_device_output(0)
    assert all(not b.is_cpu for c in got_d.columns for chunk in c.chunks for b in chunk.buffers() if b is not None), "result buffers should be kROCM"
    got = host_table(got_d).sort_by("k")
    assert got.column("k").equals(want_groups.column("k")) and got.column("v_sum").equals(want_groups.column("v_sum"))
    assert got.column("v_sum").null_count == want_groups.column("v_sum").null_count
    # a host table through the same source (whole-chunk host batches; the registered kernels take or decline them by size)
    got = plan("table_source_rocm", t, agg).to_table(use_threads=False).sort_by("k")
    assert got.column("k").equals(want_groups.column("k")) and got.column("v_sum").equals(want_groups.column("v_sum"))
    # the reference node's validation
    for bad, needle in ((lambda: acero.Declaration("table_source_rocm", acero.TableSourceNodeOptions(t), [acero.Declaration("table_source", acero.TableSourceNodeOptions(t))]).to_table(), "0 inputs"),):
        try:
            bad()
            raise SystemExit("table_source_rocm accepted an input")
        except pa.ArrowInvalid as e:
            assert needle in str(e), e
    # an empty table still produces the schema
    e = plan("table_source_rocm", t.slice(0, 0), filter_project()).to_table(use_threads=False)
    assert e.num_rows == 0 and e.schema.names == ["k", "w"], e.schema
    print("TABLE_SOURCE_OK")
''')


GOLDEN_HASH_AGGREGATE_SCRIPT = textwrap.dedent(r'''
    import ctypes, json, os, sys, faulthandler
    faulthandler.enable()
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    from tests import golden_hash_aggregate as H
    emulated = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1"
    if emulated:
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
    # the transcription holds on the reference build itself (nothing registered yet)
    assert H.replay(gold, H.stock_group_by(False)) == 28
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))     # the golden tables are tiny: send them to the GPU anyway

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def device_table(table, columns):
        # one device batch per chunk of the table (the reference tables have three), `columns` in HBM, the rest as they are
        return pa.Table.from_batches([pa.RecordBatch.from_arrays(
            [to_device(b.column(j)) if table.schema.names[j] in columns else b.column(j) for j in range(b.num_columns)],
            names=table.schema.names) for b in table.to_batches()])

    def device_values_stock(threads):
        # the STOCK GroupByNode: host keys -> the CPU Grouper -> ids; value columns in HBM -> the registered hash_* kernels
        inner = H.declaration_group_by("aggregate", threads)
        def run(table, aggs):
            cols = {c for c, _, _ in aggs if isinstance(c, str)}
            return inner(device_table(table, cols), aggs)
        return run

    def device_fused(table, aggs):
        # every column in HBM -> aggregate_rocm (the device Grouper + the dense kernels)
        specs = [(col, "hash_" + fn, H.pc_options(opts), f"out{j}") for j, (col, fn, opts) in enumerate(aggs)]
        r = acero.Declaration.from_sequence([
            acero.Declaration("table_source_rocm", acero.TableSourceNodeOptions(device_table(table, set(table.schema.names)))),
            acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions(specs, keys=["key"])),
        ]).to_table(use_threads=False)
        return r.column("key").to_pylist(), [r.column(f"out{j}").to_pylist() for j in range(len(aggs))]

    exact = [s for s in H.SECTIONS if s != "hash_mean_overflow"]
    g0, s0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1), lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    # 1. host tables under the stock GroupByNode: Table.group_by (serial and threaded/merged) and the "aggregate" node;
    #    int64 and int32 keys; MeanOverflow included (host int64 means keep the reference kernel, whatever the magnitudes)
    ran = 0
    for threads in ((False,) if emulated else (False, True)):
        ran += H.replay(gold, H.stock_group_by(threads), key_types=(pa.int64(), pa.int32()))
        ran += H.replay(gold, H.declaration_group_by("aggregate", threads))
    # 2. device-resident value columns under the stock GroupByNode (every integer type, uint64 included)
    ran += H.replay(gold, device_values_stock(False), sections=exact)
    # 3. aggregate_rocm: host batches and device-resident batches; int64 keys (the general node) and int32 keys (the fused
    #    int32 -> int64 operator takes the sum / mean / min / max / count cases, the general node the rest)
    ran += H.replay(gold, H.declaration_group_by("aggregate_rocm"), key_types=(pa.int64(), pa.int32()), sections=exact)
    ran += H.replay(gold, device_fused, key_types=(pa.int64(), pa.int32()), sections=exact)
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - g0 > 150, "the replay did not reach the HIP kernels"
    # MeanOverflow off the host route: partial sums beyond 2^53 are refused loudly, not approximated (DESIGN.md 4.6)
    for run in (device_values_stock(False), H.declaration_group_by("aggregate_rocm"), device_fused):
        try:
            H.replay(gold, run, sections=["hash_mean_overflow"])
            raise SystemExit("hash_mean beyond 2^53 on the device route did not fail")
        except pa.ArrowNotImplementedError as e:
            assert "2^53" in str(e), e
    # batches whose ARGUMENT is a scalar (CountScalar, SumMeanProductScalar, MinMaxScalar, AnyAllScalar): the registered
    # vtables' broadcast paths under the stock GroupByNode; aggregate_rocm refuses scalar columns by name
    for threads in ((False,) if emulated else (False, True)):
        ran += H.replay_scalar_arguments(gold, H.union_of_scalar_batches("aggregate", threads))
    try:
        H.replay_scalar_arguments(gold, H.union_of_scalar_batches("aggregate_rocm"))
        raise SystemExit("aggregate_rocm took a scalar column")
    except pa.ArrowNotImplementedError as e:
        assert "scalar columns" in str(e), e
    print("GOLDEN_HASH_AGGREGATE_OK", ran)
''')


GOLDEN_SCALAR_OPS_SCRIPT = textwrap.dedent(r'''
    import ctypes, json, os, sys, faulthandler
    faulthandler.enable()
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    from tests import golden_scalar_ops as S
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors_scalar.json")))
    # (cases whose operands are both scalars never reach a kernel of ours: the oracle and the stock build cover them)
    cases = list(S.cases(gold, scalar_scalar=False))
    # the reference build's own answers, before anything is registered: the transcription holds, and the bits to match
    stock = [S.check(c, lambda fn, l, r: pc.call_function(fn, [l, r])) for c in cases]
    aggs = list(S.aggregate_cases(gold, types=S.SIGNED + S.UNSIGNED))
    stock_aggs = [S.check_aggregate(c, lambda fn, x, o: pc.call_function(fn, [x], o)) for c in aggs]
    casts = list(S.cast_cases(gold))
    stock_casts = [S.check_cast(c, lambda arr, to, **o: pc.cast(arr, options=pc.CastOptions(target_type=to, **o))) for c in casts]
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))     # the golden arrays are tiny: send them to the GPU anyway

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(arr):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        arr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), arr.type)
    on_device = [0]
    def device(fn, left, right):
        # array operands in HBM (empty arrays have nothing to upload), scalars as they are; unmodified pyarrow.compute
        dl, dr = (to_device(x) if isinstance(x, pa.Array) and len(x) else x for x in (left, right))
        out = pc.call_function(fn, [dl, dr])
        if isinstance(out, pa.Array) and any(b is not None and not b.is_cpu for b in out.buffers()):
            on_device[0] += 1
            out = to_host(out)
        return out
    counters = {f: lib.arrow_amd_plugin_calls(f.encode(), 1) for f in ("greater", "compare", "add")}
    stock_before = {f: lib.arrow_amd_plugin_calls(f.encode(), 0) for f in ("greater", "compare", "add")}
    for case, want in zip(cases, stock):
        got = S.check(case, device)
        if want is not None:      # bit for bit the reference build's result (floats included: IEEE, no reassociation)
            assert S.matches(got, want, approx=False), (case["id"], case["cite"], S.as_list(got), S.as_list(want))
    ran = sum(lib.arrow_amd_plugin_calls(f.encode(), 1) - v for f, v in counters.items())
    assert len(cases) == 1950 and ran > 1500 and on_device[0] > 1400, (len(cases), ran, on_device[0])
    assert all(lib.arrow_amd_plugin_calls(f.encode(), 0) == v for f, v in stock_before.items()), "a device operand reached a reference kernel"
    # the numeric casts (scalar_cast_test.cc:269-431): values bit for bit, and the failing ones with the reference's
    # message ("Integer value V not in range: LO to HI", "Float value V was truncated converting to T")
    c0 = lib.arrow_amd_plugin_calls(b"cast", 1)
    def device_cast(arr, to, **o):
        out = pc.cast(to_device(arr), options=pc.CastOptions(target_type=to, **o))
        return to_host(out) if any(b is not None and not b.is_cpu for b in out.buffers()) else out
    for case, want in zip(casts, stock_casts):
        got = S.check_cast(case, device_cast)
        if isinstance(want, str):
            assert got == want, (case["id"], case["cite"], got, want)
    # (the counter counts completed device casts: the 20 failing cases and the zero-copy same-type ones are not in it)
    assert len(casts) == 51 and lib.arrow_amd_plugin_calls(b"cast", 1) - c0 >= 25, (len(casts), lib.arrow_amd_plugin_calls(b"cast", 1) - c0)
    # the scalar aggregates (aggregate_test.cc: SimpleSum / SimpleCount / SimpleMean / integer MinMax with their options)
    # over device-resident chunks of every integer type; an empty chunk has nothing to upload and adds nothing
    r0 = lib.arrow_amd_plugin_calls(b"reduce", 1)
    def device_aggregate(fn, chunked, options):
        chunks = [to_device(c) if len(c) else c for c in chunked.chunks]
        return pc.call_function(fn, [pa.chunked_array(chunks, chunked.type)], options)
    for case, want in zip(aggs, stock_aggs):
        got = S.check_aggregate(case, device_aggregate)
        assert got.type == want.type and (got.equals(want) or (got.as_py() != got.as_py() and want.as_py() != want.as_py())), (case["id"], got, want)
    assert len(aggs) == 8 * (21 + 15 + 25 + 18) and lib.arrow_amd_plugin_calls(b"reduce", 1) - r0 > 400, (len(aggs), lib.arrow_amd_plugin_calls(b"reduce", 1) - r0)
    # TestNumericMeanKernel.Overflow (:1349): exact on the host route (the reference kernel), refused on the device route
    mo = gold["scalar_aggregates"]["mean_overflow"]
    for name in mo["types"]:
        arr = pa.array(mo["values"], getattr(pa, name)())
        assert abs(pc.mean(arr).as_py() / mo["want"] - 1) < 1e-15
        try:
            pc.mean(to_device(arr))
            raise SystemExit("mean beyond 2^53 on device values did not fail")
        except pa.ArrowNotImplementedError as e:
            assert "2^53" in str(e), e
    # Kleene logic and invert (scalar_boolean_test.cc:54-152), every array also against each boolean scalar on either side
    b0 = lib.arrow_amd_plugin_calls(b"boolean", 1)
    nb = 0
    for fn, args, want in S.boolean_cases(gold):
        got = pc.call_function(fn, [to_device(x) if isinstance(x, pa.Array) else x for x in args])
        got = to_host(got) if any(b is not None and not b.is_cpu for b in got.buffers()) else got
        assert got.equals(want), (fn, [S.as_list(x) for x in args], S.as_list(got), S.as_list(want))
        nb += 1
    assert nb == 43 and lib.arrow_amd_plugin_calls(b"boolean", 1) - b0 == 43, (nb, lib.arrow_amd_plugin_calls(b"boolean", 1) - b0)
    # host arrays keep the reference kernels (below and above min_rows alike for these functions' tiny inputs): same answers
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(1 << 20))
    for case in cases[::7]:
        S.check(case, lambda fn, l, r: pc.call_function(fn, [l, r]))
    print("GOLDEN_SCALAR_OPS_OK", len(cases), ran, on_device[0])
''')


DEVICE_GUARD_SCRIPT = textwrap.dedent(r'''
    import ctypes, decimal, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    base = [3, 1, None, 2, 3, 0, None, 1]
    strs = ["c", "a", None, "b", "c", "", None, "a"]
    types = {"bool": pa.array([True, False, None, True, True, False, None, False]),
             **{n: pa.array(base, getattr(pa, n)()) for n in ("int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64")},
             "date32": pa.array(base, pa.int32()).cast(pa.date32()), "timestamp[us]": pa.array(base, pa.int64()).cast(pa.timestamp("us")),
             "duration[s]": pa.array(base, pa.int64()).cast(pa.duration("s")), "time32[ms]": pa.array(base, pa.int32()).cast(pa.time32("ms")),
             "string": pa.array(strs), "binary": pa.array([None if x is None else x.encode() for x in strs], pa.binary()),
             "large_string": pa.array(strs, pa.large_string()),
             "decimal128": pa.array([None if x is None else decimal.Decimal(x) for x in base], pa.decimal128(10, 2)),
             "fixed_size_binary": pa.array([None if x is None else (x + "zz")[:2].encode() for x in strs], pa.binary(2))}
    mask = pa.array([True, False, True, None, True, True, False, True])
    idx = pa.array([7, 0, None, 3, 3], pa.int32())
    # f(array, mask, indices): the functions whose kernel lists the shim extends (and the meta functions in front of them)
    fns = {"filter": lambda a, m, i: pc.filter(a, m), "take": lambda a, m, i: pc.take(a, i), "drop_null": lambda a, m, i: pc.drop_null(a),
           "unique": lambda a, m, i: pc.unique(a), "value_counts": lambda a, m, i: pc.value_counts(a),
           "dictionary_encode": lambda a, m, i: pc.dictionary_encode(a), "array_sort_indices": lambda a, m, i: pc.array_sort_indices(a),
           "sort_indices": lambda a, m, i: pc.sort_indices(a), "equal": lambda a, m, i: pc.equal(a, a), "less": lambda a, m, i: pc.less(a, a),
           "greater_scalar": lambda a, m, i: pc.greater(a, types_first[str(a.type)]), "add": lambda a, m, i: pc.add(a, a),
           "subtract_checked": lambda a, m, i: pc.subtract_checked(a, a), "multiply": lambda a, m, i: pc.multiply(a, a),
           "indices_nonzero": lambda a, m, i: pc.indices_nonzero(a), "count": lambda a, m, i: pc.count(a),
           "count_null": lambda a, m, i: pc.count(a, mode="only_null"), "is_null": lambda a, m, i: pc.is_null(a),
           "is_valid": lambda a, m, i: pc.is_valid(a)}
    types_first = {str(a.type): a[0] for a in types.values()}
    def run(f, a, m, i):
        try:
            return ("ok", f(a, m, i))
        except (pa.ArrowInvalid, pa.ArrowNotImplementedError, pa.ArrowTypeError) as e:
            return ("err", type(e).__name__ + ": " + str(e))
    # the reference build on host arrays, before anything is registered
    before = {(fn, tn): run(f, arr, mask, idx) for fn, f in fns.items() for tn, arr in types.items()}
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    # 1. host arrays: the same results and the same errors with the guards in front of the reference kernels, whether the
    #    shim's size threshold sends them to the GPU or not
    for min_rows in (1 << 40, 0):
        lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(min_rows))
        for (fn, tn), (kind, want) in before.items():
            k2, got = run(fns[fn], types[tn], mask, idx)
            same = kind == k2 and (want == got if kind == "err" else want.equals(got))
            assert same, (min_rows, fn, tn, kind, str(want)[:200], k2, str(got)[:200])

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    # 2. device-resident arrays: computed on the device where a device kernel exists, REFUSED by name everywhere else —
    #    never handed to a CPU kernel (which would read Buffer::data() == nullptr or an HBM address)
    d_mask, d_idx = to_device(mask), to_device(idx)
    done = refused = 0
    for fn, f in fns.items():
        for tn, arr in types.items():
            kind, got = run(f, to_device(arr), d_mask, d_idx)
            if kind == "ok":
                assert before[(fn, tn)][0] == "ok", (fn, tn)
                done += 1
            else:
                host_kind, host_err = before[(fn, tn)]
                assert host_kind == "err" or "arrow_amd" in got, (fn, tn, got)      # (errors the reference raises for host arrays too are fine)
                if "no device kernel is registered" in got:      # (the guard names the type it refused)
                    assert tn.split("[")[0] in got or str(arr.type) in got, (fn, tn, got)
                refused += 1
    assert done > 150 and refused > 60, (done, refused)
    # 3. is_valid / is_null / true_unless_null (plugin/validity.inc): the reference's kernels see no validity bitmap in a
    #    device-resident array and answer "no nulls" — silently; the twins answer from the bitmap in HBM, result in HBM
    def to_host(arr):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        arr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), arr.type)
    rng = np.random.default_rng(3)
    m = 70_001
    big = pa.array(rng.integers(-9, 9, m), mask=rng.random(m) < 0.3)
    vcases = dict(types, big=big, no_nulls=pa.array(np.arange(1000)), f64=pa.array([1.0, float("nan"), None, 2.0]))
    b0 = lib.arrow_amd_plugin_calls(b"boolean", 1)
    for name, arr in vcases.items():
        d = to_device(arr)
        for fn in ("is_valid", "is_null", "true_unless_null"):
            for host, dev in ((arr, d), (arr.slice(3, max(len(arr) - 4, 1)), d.slice(3, max(len(arr) - 4, 1)))):
                got = pc.call_function(fn, [dev])
                assert all(not b.is_cpu for b in got.buffers() if b is not None), (fn, name, "the result should stay in HBM")
                assert to_host(got).equals(pc.call_function(fn, [host])), (fn, name)
    assert lib.arrow_amd_plugin_calls(b"boolean", 1) - b0 == 6 * len(vcases)
    # `count` has the same blind spot in the reference (length - GetNullCount() of a span without a validity pointer):
    # every type with a physical validity bitmap is counted from that bitmap in HBM
    for name, arr in vcases.items():
        d = to_device(arr)
        for mode in ("only_valid", "only_null", "all"):
            assert pc.count(d, mode=mode).equals(pc.count(arr, mode=mode)), (name, mode)
            assert pc.count(d.slice(3, max(len(arr) - 4, 1)), mode=mode).equals(pc.count(arr.slice(3, max(len(arr) - 4, 1)), mode=mode)), (name, mode)
        # (chunks uploaded separately, offsets 0: pyarrow's ChunkedArray constructor counts the nulls of a SLICED array on the CPU)
        halves = [pa.concat_arrays([arr.slice(0, len(arr) // 2)]), pa.concat_arrays([arr.slice(len(arr) // 2)])]
        chunked = pa.chunked_array([to_device(h) for h in halves])
        assert pc.count(chunked).equals(pc.count(arr)), name
    # a plan over a device-resident table: filter(is_valid(v) and not is_null(k)) keeps exactly the rows the host plan keeps
    from pyarrow import acero
    tab = pa.table({"v": big, "k": pa.array(rng.integers(0, 5, m), mask=rng.random(m) < 0.1), "x": pa.array(rng.random(m))})
    dtab = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(tab.column(j).chunk(0)) for j in range(3)], names=tab.schema.names)])
    pred = pc.field("v").is_valid() & ~pc.field("k").is_null()
    def plan(source, t):
        return acero.Declaration.from_sequence([acero.Declaration(source, acero.TableSourceNodeOptions(t)),
                                                acero.Declaration("filter", acero.FilterNodeOptions(pred))]).to_table(use_threads=False)
    want = plan("table_source", tab)
    got = plan("table_source_rocm", dtab)
    got = pa.table({n: pa.chunked_array([to_host(c) for c in got.column(n).chunks], got.schema.field(n).type) for n in got.schema.names})
    assert 0 < want.num_rows < m and got.equals(want), (got.num_rows, want.num_rows)
    # refused by name on the device route: nulls that are not the validity bitmap, NaN as null
    for arr, kw, needle in ((pa.RunEndEncodedArray.from_arrays(pa.array([2, 3], pa.int32()), pa.array([1, None], pa.int64())), {}, "not its validity bitmap"),
                            (pa.array([1.0, float("nan"), None]), {"nan_is_null": True}, "nan_is_null")):
        try:
            pc.is_null(to_device(arr), **kw)
            raise SystemExit("is_null took " + str(arr.type) + str(kw))
        except pa.ArrowNotImplementedError as e:
            assert needle in str(e), e
    print("DEVICE_GUARD_OK", done, refused)
''')


REE_FILTER_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(91)
    n = SC(1_000_000)
    # runs of 1..40 rows, ~40 % selected, 10 % of the RUN VALUES null
    lens = rng.integers(1, 41, n)
    ends = np.cumsum(lens)
    runs = int(np.searchsorted(ends, n)) + 1
    ends = ends[:runs].copy(); ends[-1] = n
    run_vals = pa.array(rng.random(runs) < 0.4, mask=rng.random(runs) < 0.1)
    cases = []
    for end_type in (pa.int16(), pa.int32(), pa.int64()):
        m = n if end_type != pa.int16() else min(n, 30_000)
        r = int(np.searchsorted(ends, m)) + 1
        e = ends[:r].copy(); e[-1] = m
        ree = pa.RunEndEncodedArray.from_arrays(pa.array(e, end_type), run_vals.slice(0, r))
        vals = pa.array(rng.integers(-2**62, 2**62, m), mask=rng.random(m) < 0.05)
        small = pa.array(rng.integers(-100, 100, m).astype(np.int16))
        dbl = pa.array(rng.standard_normal(m))
        for v in (vals, small, dbl):
            for mode in ("drop", "emit_null"):
                cases.append((v, ree, mode, pc.filter(v, ree, null_selection_behavior=mode)))
                # a logical slice of both (the REE array keeps its runs and gets an offset)
                o, l = m // 7 + 3, m // 2
                cases.append((v.slice(o, l), ree.slice(o, l), mode, pc.filter(v.slice(o, l), ree.slice(o, l), null_selection_behavior=mode)))
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(k) for k in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(arr):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        arr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), arr.type)
    g0, s0 = lib.arrow_amd_plugin_calls(b"array_filter", 1), lib.arrow_amd_plugin_calls(b"array_filter", 0)
    for v, ree, mode, want in cases:
        got = pc.call_function("array_filter", [to_device(v), to_device(ree)], pc.FilterOptions(null_selection_behavior=mode))
        got = to_host(got)
        assert got.type == want.type and len(got) == len(want), (got.type, len(got), len(want))
        assert got.equals(want), (str(v.type), str(ree.type), mode, v.offset)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) - g0 == len(cases), "the REE filters did not run on the device"
    # the reference's own filter vectors (vector_selection_test.cc:319-336), which its harness also runs with the filter
    # run-end encoded (:88,134,310): here on device arrays, EMIT_NULL and DROP (= the emitted nulls removed)
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
    lib.arrow_amd_plugin_set_min_rows.argtypes = [ctypes.c_int64]
    lib.arrow_amd_plugin_set_min_rows(0)
    ran = 0
    for case in gold["filter_emit_null"] + [dict(gold["filter_sliced_mask"], sliced=True)]:
        vals = pa.array(case["values"], pa.int32())
        if case.get("sliced"):
            mask = pa.array(case["mask_full"], pa.bool_()).slice(case["mask_offset"], case["mask_length"])
        else:
            mask = pa.array(case["mask"], pa.bool_())
        if len(vals) == 0:
            continue
        for end_type in ("int16", "int32", "int64"):
            ree = pc.run_end_encode(mask, run_end_type=end_type)
            got = to_host(pc.call_function("array_filter", [to_device(vals), to_device(ree)],
                                           pc.FilterOptions(null_selection_behavior="emit_null")))
            assert got.to_pylist() == case["want"], (case["cite"], end_type, got.to_pylist())
            ran += 1
    assert ran >= 30, ran
    lib.arrow_amd_plugin_set_min_rows(1 << 16)
    # host arrays with an REE filter keep the reference kernel (same exec serves both layouts)
    v, ree, mode, want = cases[0]
    assert pc.filter(v, ree, null_selection_behavior=mode).equals(want)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 0) > s0
    # mixed residency is refused by name
    try:
        pc.call_function("array_filter", [to_device(v), ree], pc.FilterOptions())
        raise SystemExit("mixed residency accepted")
    except pa.ArrowNotImplementedError as e:
        assert "both be device-resident" in str(e), e
    print("REE_FILTER_OK")
''')


FLOAT_EXTREMA_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        pa.set_cpu_count(1)
        pa.set_io_thread_count(1)
    # hash_min / hash_max / hash_min_max of float32 / float64 and the temporal types (VERDICT r3 "missing" 4): MinMaxOp =
    # fmin / fmax over NaN anti-extrema (kernels/hash_aggregate.cc:306-326) — NaN rows are skipped, a group of NaNs only
    # ends as NaN, +-inf are ordinary values; +0.0 and -0.0 compare equal (fmin leaves that tie to the row order).
    rng = np.random.default_rng(41)
    n = SC(600_000)
    G = 300
    k = pa.array(rng.integers(0, G, n), mask=rng.random(n) < 0.01)
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)
    x[rng.random(n) < 0.05] = np.nan
    x[rng.random(n) < 0.02] = np.inf
    x[rng.random(n) < 0.02] = -np.inf
    x[rng.random(n) < 0.05] = 0.0
    x[rng.random(n) < 0.05] = -0.0
    kk = np.asarray(k.fill_null(0))
    x[kk == 7] = np.nan                      # a group of NaNs only
    x[kk == 8] = np.where(rng.random((kk == 8).sum()) < 0.5, 0.0, -0.0)   # a group of zeros of both signs
    fmask = rng.random(n) < 0.2
    fmask[kk == 9] = True                    # a group of nulls only
    f64 = pa.array(x, mask=fmask)
    y = (rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n)).astype(np.float32)
    y[rng.random(n) < 0.05] = np.nan
    f32 = pa.array(y, mask=rng.random(n) < 0.2)
    ts = pa.array(rng.integers(-2**60, 2**60, n), pa.timestamp("ns", tz="UTC"), mask=rng.random(n) < 0.2)
    d32 = pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), pa.date32(), mask=rng.random(n) < 0.2)
    t64 = pa.array(rng.integers(0, 86_400_000_000, n), pa.time64("us"), mask=rng.random(n) < 0.2)
    t32 = pa.array(rng.integers(0, 86_400, n).astype(np.int32), pa.time32("s"), mask=rng.random(n) < 0.2)
    d64 = pa.array(rng.integers(-10**6, 10**6, n) * 86_400_000, pa.date64(), mask=rng.random(n) < 0.2)
    i64 = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.2)
    t = pa.table({"k": k, "f64": f64, "f32": f32, "ts": ts, "d32": d32, "t64": t64, "t32": t32, "d64": d64, "i64": i64})
    tc = pa.concat_tables([t.slice(0, n // 3), t.slice(n // 3, n // 5), t.slice(n // 3 + n // 5)])
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    cols = ["f64", "f32", "ts", "d32", "t64", "t32", "d64", "i64"]     # (duration: no kernel in the reference)
    aggs = [(c, fn, o) for c in cols for fn in ("min", "max", "min_max") for o in (None, strict)]

    def run(tab, threads):
        return tab.group_by("k", use_threads=threads).aggregate(aggs).sort_by("k")

    def same(a, b, what):
        a, b = ((pa.concat_arrays(z.chunks) if z.num_chunks else pa.array([], z.type)) if isinstance(z, pa.ChunkedArray) else z
                for z in (a, b))
        assert a.type == b.type, (what, a.type, b.type)
        if pa.types.is_struct(a.type):
            for i in range(a.type.num_fields):
                same(a.field(i), b.field(i), (what, a.type.field(i).name))
            return
        assert a.is_null().equals(b.is_null()), (what, "validity", a.null_count, b.null_count)
        if pa.types.is_floating(a.type):
            an, bn = pc.is_nan(a).fill_null(False), pc.is_nan(b).fill_null(False)
            assert an.equals(bn), (what, "NaN groups differ")
            a, b = pc.if_else(an, 0.0, a), pc.if_else(bn, 0.0, b)
        assert a.equals(b), (what, a.slice(0, 8), b.slice(0, 8))

    want = {(name, threads): run(tab, threads) for name, tab in (("t", t), ("tc", tc)) for threads in (False, True)}
    w0 = want[("t", False)]
    nan_group = w0.column("k").to_pylist().index(7)
    c0 = w0.schema.names.index("f64_min")
    assert np.isnan(w0.column(c0)[nan_group].as_py()) and w0.column(c0)[nan_group].is_valid   # the reference: NaN, valid
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    gpu0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    for (name, threads), w in want.items():
        got = run(t if name == "t" else tc, threads)
        assert got.schema.equals(w.schema), (got.schema, w.schema)
        for col in range(w.num_columns):
            same(got.column(col), w.column(col), (name, threads, w.schema.names[col], col))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - gpu0 >= 4 * len(aggs), "the extrema vtables did not run on the device"

    # ---- device-resident value columns under the stock GroupByNode, and the aggregate_rocm node
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    m = SC(200_000)
    th = t.slice(0, m).combine_chunks()
    td = pa.table({"k": th.column("k").chunk(0), **{c: to_device(th.column(c).chunk(0)) for c in cols}})
    daggs = [(c, "hash_" + fn, o, "%s_%s_%d" % (c, fn, o is strict)) for c in cols for fn in ("min", "max", "min_max") for o in (None, strict)]
    def plan(tab, node="aggregate", dg=daggs):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(dg, keys=["k"]))]).to_table(use_threads=False).sort_by("k")
    wh = plan(th)       # the host route, shown equal to the reference above
    stock2 = lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    gd = plan(td)
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock2, "device-resident values must not reach a reference kernel"
    for col in wh.schema.names:
        same(gd.column(col), wh.column(col), ("device values", col))
    rocm_aggs = [a for a in daggs if "min_max" not in a[1]]
    wr = plan(th, "aggregate", rocm_aggs)
    for tab, what in ((th, "aggregate_rocm host"), (td, "aggregate_rocm device")):
        gr = plan(tab, "aggregate_rocm", rocm_aggs)
        assert gr.schema.equals(wr.schema), (gr.schema, wr.schema)
        for col in wr.schema.names:
            same(gr.column(col), wr.column(col), (what, col))
    print("FLOAT_EXTREMA_OK")
""")


FLOAT_AGGREGATE_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # VERDICT r3 missing 5: scalar sum / mean / min_max / min / max of float32 / float64 / boolean / temporal device columns.
    # The float SUM is the reference's bit for bit (SumArray's pairwise summation tree evaluated on the device,
    # arx_sum_float); extrema by order keys (NaNs skipped, a column of NaNs ends as NaN); booleans from two popcounts.
    rng = np.random.default_rng(53)
    n = SC(2_000_003)
    def floats(dtype, m, null_p):
        x = (rng.standard_normal(m) * 10.0 ** rng.integers(-12, 12, m)).astype(dtype)
        x[rng.random(m) < 0.01] = 0.0
        x[rng.random(m) < 0.01] = -0.0
        return pa.array(x, mask=(rng.random(m) < null_p) if null_p else None)
    cols = {
        "f64": floats(np.float64, n, 0.1), "f64_dense": floats(np.float64, n, 0.0), "f64_sparse": floats(np.float64, n // 4, 0.95),
        "f32": floats(np.float32, n, 0.1), "f32_dense": floats(np.float32, n // 2, 0.0),
        "f64_inf": pa.array([1.0, float("inf"), None, -2.5, float("inf")] * 50), "f64_nan": pa.array([float("nan"), None, float("nan")] * 40),
        "f64_mixed_nan": pa.array([float("nan"), 3.0, None, -7.0, float("nan"), 0.5] * 30),
        "f64_null": pa.array([None] * 500, pa.float64()), "f64_empty": pa.array([], pa.float64()),
        "f32_one": pa.array([1.25], pa.float32()),
        "b": pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.1), "b_dense": pa.array(rng.random(n // 3) < 0.999),
        "b_true": pa.array([True, None, True] * 70), "b_false": pa.array([False] * 130), "b_null": pa.array([None] * 65, pa.bool_()),
        "ts": pa.array(rng.integers(-2**60, 2**60, n // 2), pa.timestamp("ns", tz="UTC"), mask=rng.random(n // 2) < 0.1),
        "d32": pa.array(rng.integers(-2**31, 2**31, n // 2).astype(np.int32), pa.date32(), mask=rng.random(n // 2) < 0.1),
        "d64": pa.array(rng.integers(-10**6, 10**6, n // 3) * 86_400_000, pa.date64()),
        "t32": pa.array(rng.integers(0, 86_400_000, n // 3).astype(np.int32), pa.time32("ms"), mask=rng.random(n // 3) < 0.3),
        "t64": pa.array(rng.integers(0, 86_400_000_000_000, n // 3), pa.time64("ns"), mask=rng.random(n // 3) < 0.3),
        "s": pa.array(["pear", None, "apple", "zebra", ""] * 11),
    }
    optss = (None, pc.ScalarAggregateOptions(skip_nulls=False), pc.ScalarAggregateOptions(min_count=0),
             pc.ScalarAggregateOptions(skip_nulls=True, min_count=10**9))
    def fns_of(a):
        if pa.types.is_floating(a.type) or pa.types.is_boolean(a.type):
            return ("sum", "mean", "min_max", "min", "max")
        return ("min_max", "min", "max")
    def views(a):     # the column, a slice at an odd offset, two chunks (merge of two states)
        out = [("whole", a)]
        if len(a) > 200:
            out.append(("slice", a.slice(37, len(a) - 100)))
            out.append(("chunks", pa.chunked_array([a.slice(0, len(a) // 3 + 5), a.slice(len(a) // 3 + 5)])))
        return out
    def bits(x):      # scalars compared exactly: doubles by their bit patterns (NaN == NaN, -0.0 != 0.0 would show)
        if isinstance(x, pa.StructScalar):
            return tuple(bits(v) for v in x.values())
        if not x.is_valid:
            return (str(x.type), None)
        if pa.types.is_temporal(x.type):
            return (str(x.type), x.value)
        v = x.as_py()
        if isinstance(v, float):
            return (str(x.type), "nan" if v != v else np.float64(v).tobytes())
        return (str(x.type), v)
    want = {}
    for name, a in cols.items():
        for vname, va in views(a):
            for oi, o in enumerate(optss):
                for fn in fns_of(a):
                    want[(name, vname, oi, fn)] = bits(pc.call_function(fn, [va], o))
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def dev_view(va):
        if isinstance(va, pa.ChunkedArray):
            return pa.chunked_array([to_device(c) for c in va.chunks])
        return to_device(va)
    red0, stock0 = lib.arrow_amd_plugin_calls(b"reduce", 1), lib.arrow_amd_plugin_calls(b"reduce", 0)
    zero_ties = 0
    for name, a in cols.items():
        for vname, va in views(a):
            dv = dev_view(va) if name != "s" else None
            for oi, o in enumerate(optss):
                for fn in fns_of(a):
                    w = want[(name, vname, oi, fn)]
                    h = bits(pc.call_function(fn, [va], o))          # host batches through the plugged registry: the reference kernel
                    assert h == w, ("host", name, vname, oi, fn, h, w)
                    if dv is None:
                        continue
                    g = bits(pc.call_function(fn, [dv], o))
                    if g != w and fn in ("min_max", "min", "max") and pa.types.is_floating(a.type):
                        # the one tie fmin / fmax leave to the row order: an extremum of 0.0 where zeros of both signs occur
                        gz = pc.call_function(fn, [dv], o)
                        wz = pc.call_function(fn, [va], o)
                        flat = lambda z: [v.as_py() for v in z.values()] if isinstance(z, pa.StructScalar) else [z.as_py()]
                        assert all(x == y for x, y in zip(flat(gz), flat(wz))), ("device", name, vname, oi, fn, gz, wz)
                        zero_ties += 1
                        continue
                    assert g == w, ("device", name, vname, oi, fn, g, w)
    assert lib.arrow_amd_plugin_calls(b"reduce", 1) - red0 > 400, "the device aggregates did not run"
    assert lib.arrow_amd_plugin_calls(b"reduce", 0) > stock0
    # what has no device kernel is refused by name (and its host route is the reference's, shown above for the strings)
    ds = to_device(cols["s"])
    for fn in ("min_max", "min", "max"):
        try:
            pc.call_function(fn, [ds])
            raise SystemExit("expected NotImplemented for " + fn + " of device strings")
        except pa.lib.ArrowNotImplementedError as e:
            assert "device-resident" in str(e), e
    import decimal
    # decimal128 is served since late round 4 (DECIMAL_SUM_SCRIPT has the cases); decimal256 has no device form: refused by name
    hd = pa.array([decimal.Decimal("1.5"), None, decimal.Decimal("-2.25")], pa.decimal128(10, 2))
    dd = to_device(hd)
    for fn in ("sum", "mean", "min_max"):
        assert pc.call_function(fn, [dd]).equals(pc.call_function(fn, [hd])), fn
    d256 = to_device(pa.array([decimal.Decimal("1.5"), None], pa.decimal256(40, 2)))
    for fn in ("sum", "mean", "min_max"):
        try:
            pc.call_function(fn, [d256])
            raise SystemExit("expected NotImplemented for " + fn + " of device decimal256")
        except pa.lib.ArrowNotImplementedError as e:
            assert "device-resident" in str(e), e
    try:
        pc.sum(pa.chunked_array([cols["f64"].slice(0, 10), to_device(cols["f64"].slice(10, 50))]))
        raise SystemExit("expected NotImplemented for a host+device aggregation")
    except pa.lib.ArrowNotImplementedError:
        pass
    print("FLOAT_AGGREGATE_OK", zero_ties)
""")


FILL_NULL_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # VERDICT r3 missing 5: fill_null on device-resident arrays.  pyarrow's fill_null(values, fill) is coalesce(values, fill)
    # (CoalesceFunctor, kernels/scalar_if_else.cc); the results below are computed by the reference BEFORE the plugin is loaded.
    rng = np.random.default_rng(61)
    n = SC(1_000_003)
    def col(t, null_p):
        mask = (rng.random(n) < null_p) if null_p else None
        if pa.types.is_boolean(t):
            return pa.array(rng.random(n) < 0.5, mask=mask)
        if pa.types.is_floating(t):
            return pa.array(rng.standard_normal(n).astype(t.to_pandas_dtype()), mask=mask)
        bits = t.bit_width
        raw = rng.integers(0, 2**(bits - 1) - 1, n).astype("int%d" % bits)
        if pa.types.is_time(t):
            raw = raw % (86_400 if t.unit == "s" else 86_400_000_000)
        return pa.array(raw, pa.int64() if bits == 64 else pa.int32() if bits == 32 else pa.int16() if bits == 16 else pa.int8(), mask=mask).cast(t) \
            if not pa.types.is_unsigned_integer(t) else pa.array(raw.astype("uint%d" % bits), t, mask=mask)
    types = [pa.bool_(), pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64(), pa.float32(),
             pa.float64(), pa.date32(), pa.date64(), pa.time32("s"), pa.time64("us"), pa.timestamp("ns", tz="UTC"), pa.duration("ms")]
    cases, want = [], {}
    for t in types:
        a, b, dense = col(t, 0.3), col(t, 0.2), col(t, 0.0)
        fill = b[int(np.flatnonzero(np.asarray(b.is_valid()))[0])]
        for name, x, y in (("scalar", a, fill), ("null_scalar", a, pa.scalar(None, t)), ("array", a, b), ("no_nulls", dense, fill),
                           ("slices", a.slice(13, n - 100), b.slice(29, n - 100)), ("empty", a.slice(0, 0), fill)):
            cases.append((str(t), name, x, y))
            want[(str(t), name)] = pc.fill_null(x, y)
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(x):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    gpu0 = lib.arrow_amd_plugin_calls(b"coalesce", 1)
    for tname, name, x, y in cases:
        w = want[(tname, name)]
        h = pc.fill_null(x, y)                        # host operands through the plugged registry: the reference kernel
        assert h.equals(w) and h.type == w.type, ("host", tname, name)
        dx = to_device(x)
        dy = to_device(y) if isinstance(y, pa.Array) else y
        g = pc.fill_null(dx, dy)
        assert g.type == w.type and len(g) == len(w), ("device", tname, name, g.type, w.type)
        if len(g):
            assert not g.buffers()[1].is_cpu, ("the result of a device fill_null lives in HBM", tname, name)
        gh = to_host(g)
        assert gh.equals(w), ("device", tname, name, gh.slice(0, 8), w.slice(0, 8))
        assert gh.null_count == w.null_count, ("null_count", tname, name, gh.null_count, w.null_count)
    assert lib.arrow_amd_plugin_calls(b"coalesce", 1) - gpu0 >= sum(1 for c in cases if len(c[2])), "coalesce did not run on the device"
    # what has no device kernel is refused by name, not handed HBM pointers
    for bad in (pa.array(["a", None, "c"]), pa.array([b"xy", None], pa.binary(2))):
        try:
            pc.fill_null(to_device(bad), bad[0])
            raise SystemExit("expected NotImplemented for fill_null of device " + str(bad.type))
        except pa.lib.ArrowNotImplementedError as e:
            assert "device" in str(e), e
    da = to_device(col(pa.int64(), 0.3))
    try:
        pc.coalesce(da, da, da)
        raise SystemExit("expected NotImplemented for a three-operand device coalesce")
    except pa.lib.ArrowNotImplementedError as e:
        assert "coalesce" in str(e), e
    assert pc.coalesce(pa.array([None, 1, None]), pa.array([None, 5, 7]), pa.array([9, 9, 9])).to_pylist() == [9, 1, 7]   # host varargs: the reference
    print("FILL_NULL_OK")
""")


WRAP_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    EMULATED = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1"
    if EMULATED:
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.arrow_amd_wrap_device_memory.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))
    # arrow_amd_wrap_device_memory: memory the caller owns (a torch tensor on the GPU; a numpy array under the emulated HIP
    # runtime, whose "device" memory is host memory) becomes a device-resident pyarrow array without a copy — the route
    # bench.py takes from its generated HBM buffers to pyarrow.compute
    if EMULATED:
        keep = []
        def device_memory(a):
            a = np.ascontiguousarray(a)
            keep.append(a)
            return a.ctypes.data
    else:
        import torch
        keep = []
        def device_memory(a):
            t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).cuda()
            keep.append(t)
            return t.data_ptr()
    def wrap(arr):
        bufs = arr.buffers()
        assert arr.offset == 0
        nbytes = lambda b: np.frombuffer(b, np.uint8)
        vptr = device_memory(np.concatenate([nbytes(bufs[0]), np.zeros(8, np.uint8)])) if bufs[0] is not None else None
        dptr = device_memory(np.concatenate([nbytes(bufs[1]), np.zeros(8, np.uint8)]))
        c_schema, c_dev = ctypes.create_string_buffer(72), ctypes.create_string_buffer(128)
        arr.type._export_to_c(ctypes.addressof(c_schema))
        assert lib.arrow_amd_wrap_device_memory(ctypes.addressof(c_schema), len(arr), -1, vptr, dptr, ctypes.addressof(c_dev)) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(x):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    rng = np.random.default_rng(71)
    n = SC(1_000_003)
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    dense = pa.array(rng.standard_normal(n))
    mask = pa.array(rng.random(n) < 0.1)
    flags = pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2)
    dv, dd, dm, df = wrap(vals), wrap(dense), wrap(mask), wrap(flags)
    for d, h in ((dv, vals), (dd, dense), (dm, mask), (df, flags)):
        assert not d.buffers()[1].is_cpu and len(d) == len(h)
        assert to_host(d).equals(h) and to_host(d).null_count == h.null_count
    g0 = lib.arrow_amd_plugin_calls(b"array_filter", 1)
    out = pc.filter(dv, dm)
    rows = pc.indices_nonzero(dm)
    tk = pc.take(dv, rows, boundscheck=False)
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) > g0
    assert rows.type == pa.uint64() and to_host(rows).equals(pc.indices_nonzero(mask))
    assert to_host(out).equals(pc.filter(vals, mask)) and to_host(tk).equals(pc.filter(vals, mask))
    assert pc.sum(dd).equals(pc.sum(dense)) and pc.min_max(dv).equals(pc.min_max(vals))
    assert to_host(pc.fill_null(df, False)).equals(pc.fill_null(flags, False))
    assert to_host(pc.indices_nonzero(df)).equals(pc.indices_nonzero(flags))     # (nulls are not "non-zero")
    print("WRAP_OK")
""")


ACERO_OVERRIDE_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # VERDICT r3 weak 9: an UNMODIFIED plan — table_source / filter / project / aggregate / order_by by their stock names —
    # over a device-resident table lands on the plugin's nodes once arrow_amd_override_acero_factories(1) was called.
    rng = np.random.default_rng(91)
    n = SC(4_000_000)
    k = pa.array(rng.integers(0, 5000, n).astype(np.int32), mask=rng.random(n) < 0.01)
    v = pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.1)
    x = pa.array(rng.random(n))
    host = pa.table({"k": k, "v": v, "x": x})
    def group_plan(tab, aggregate="aggregate"):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("filter", acero.FilterNodeOptions(pc.field("x") > 0.25)),
            acero.Declaration("project", acero.ProjectNodeOptions([pc.field("k"), pc.add(pc.field("v"), pc.field("v"))], ["k", "v2"])),
            acero.Declaration(aggregate, acero.AggregateNodeOptions([("v2", "hash_sum", None, "s"), ("v2", "hash_count", None, "c")], keys=["k"]))])
    def scalar_plan(tab):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("filter", acero.FilterNodeOptions(pc.field("x") > 0.25)),
            acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "sum", None, "s"), ("v", "min_max", None, "mm"), ("x", "max", None, "xm")]))])
    def order_plan(tab):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("order_by", acero.OrderByNodeOptions([("v", "descending"), ("k", "ascending")], null_placement="at_start"))])
    want_group = group_plan(host).to_table(use_threads=False).sort_by("k")
    want_scalar = scalar_plan(host).to_table(use_threads=False)
    want_order = order_plan(host.slice(0, n // 8)).to_table(use_threads=False)
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def dev_table(tab):
        return pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(tab.column(j).combine_chunks()) if tab.column(j).num_chunks != 1 else to_device(tab.column(j).chunk(0))
                                                                  for j in range(tab.num_columns)], names=tab.schema.names)])
    dev = dev_table(host)
    def to_host(x):
        if all(b is None or b.is_cpu for b in x.buffers()):
            return x
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    def host_table(t):      # (order_by_rocm leaves its result in HBM)
        return pa.table({name: pa.chunked_array([to_host(c) for c in t.column(name).chunks], t.schema.field(name).type) for name in t.schema.names})
    def filter_calls():
        return lib.arrow_amd_plugin_calls(b"array_filter", 1)
    morsels = (n + 32767) // 32768
    # ---- as arrow_amd_register() leaves it (round 6, VERDICT r5 weak 8): a stock `table_source` over a device-resident table
    # delivers WHOLE CHUNKS — one filter call per column, not one per 32Ki-row morsel and column — with nothing but the
    # registration called; the keyed `aggregate` over device-resident key columns is the guard's (-> aggregate_rocm).
    f0 = filter_calls()
    assert group_plan(dev, "aggregate_rocm").to_table(use_threads=False).sort_by("k").equals(want_group)
    assert filter_calls() - f0 == 3, ("a device table source must deliver whole chunks by default", filter_calls() - f0)
    f0 = filter_calls()
    assert group_plan(dev).to_table(use_threads=False).sort_by("k").equals(want_group)
    assert filter_calls() - f0 == 3, filter_calls() - f0
    assert scalar_plan(dev).to_table(use_threads=False).equals(want_scalar)
    # ---- the opt-out (-1): the reference SourceNode's 32Ki-row morsels — one filter call per morsel and column
    assert lib.arrow_amd_override_acero_factories(-1) == 0, lib.arrow_amd_plugin_last_error()
    f0 = filter_calls()
    assert group_plan(dev, "aggregate_rocm").to_table(use_threads=False).sort_by("k").equals(want_group)
    assert filter_calls() - f0 >= 3 * morsels, "with the opt-out the stock source cuts the table into 32Ki-row morsels"
    # ---- on
    assert lib.arrow_amd_override_acero_factories(1) == 0, lib.arrow_amd_plugin_last_error()
    assert lib.arrow_amd_override_acero_factories(1) == 0      # (idempotent)
    f0 = filter_calls()
    got = group_plan(dev).to_table(use_threads=False).sort_by("k")
    assert got.equals(want_group), (got.slice(0, 5), want_group.slice(0, 5))
    assert filter_calls() - f0 == 3, ("the device table went through the stock source", filter_calls() - f0)   # one call per column, whole chunk
    assert group_plan(dev).to_table(use_threads=True).sort_by("k").equals(want_group)
    # no keys: aggregate_rocm declines, the wrapper hands the stock ScalarAggregateNode the same whole-chunk batches
    got = scalar_plan(dev).to_table(use_threads=False)
    assert got.equals(want_scalar), (got, want_scalar)
    got = host_table(order_plan(dev_table(host.slice(0, n // 8))).to_table(use_threads=False))
    assert got.equals(want_order)
    # host tables are none of the wrappers' business
    assert group_plan(host).to_table(use_threads=False).sort_by("k").equals(want_group)
    assert order_plan(host.slice(0, n // 8)).to_table(use_threads=False).equals(want_order)
    # ---- 0 again: the state of arrow_amd_register() (guard + whole-chunk device sources); order_by is the stock node's again
    assert lib.arrow_amd_override_acero_factories(0) == 0, lib.arrow_amd_plugin_last_error()
    f0 = filter_calls()
    assert group_plan(dev, "aggregate_rocm").to_table(use_threads=False).sort_by("k").equals(want_group)
    assert filter_calls() - f0 == 3
    # a plan that ended leaves nothing behind in the guard's bookkeeping: many plans over short-lived device tables
    for _ in range(40):
        t = dev_table(host.slice(0, 1000))
        assert group_plan(t).to_table(use_threads=False).num_rows > 0
        del t
    print("ACERO_OVERRIDE_OK")
""")


LARGE_BINARY_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # VERDICT r3 missing 6: filter / take of large_utf8 / large_binary (int64 offsets) on device-resident arrays — the
    # reference runs all four base-binary types through one VarBinary implementation
    # (vector_selection_filter_internal.cc:835-848, vector_selection_take_internal.cc)
    rng = np.random.default_rng(97)
    n = SC(500_000)
    words = ["", "a", "bc", "def" * 7, "été", "x" * 100, "0123456789" * 30, "z\x00z"]
    pick = rng.integers(0, len(words), n)
    cols = {
        "large_utf8": pa.array([words[i] for i in pick], pa.large_utf8(), mask=rng.random(n) < 0.1),
        "large_binary": pa.array([words[i].encode() * (i % 3) for i in pick], pa.large_binary(), mask=rng.random(n) < 0.05),
        "large_utf8_dense": pa.array([words[i] for i in pick[: n // 2]], pa.large_utf8()),
        "utf8": pa.array([words[i] for i in pick], pa.utf8(), mask=rng.random(n) < 0.1),
    }
    mask = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.05)
    idx = pa.array(rng.integers(0, n // 2, n // 3), pa.int64(), mask=rng.random(n // 3) < 0.1)
    idx32 = pa.array(rng.integers(0, n // 2, 1000).astype(np.uint32))
    want = {}
    for name, a in cols.items():
        m = mask.slice(0, len(a))
        want[name] = (pc.filter(a, m), pc.filter(a, m, null_selection_behavior="emit_null"), pc.take(a, idx), pc.take(a, idx32),
                      pc.filter(a.slice(11, len(a) - 50), m.slice(11, len(a) - 50)), pc.take(a.slice(7), idx32), pc.drop_null(a))
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(x):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    g0 = lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1)
    dmask, didx, didx32 = to_device(mask), to_device(idx), to_device(idx32)
    for name, a in cols.items():
        d = to_device(a)
        m, dm = mask.slice(0, len(a)), dmask.slice(0, len(a))
        got = (pc.filter(d, dm), pc.filter(d, dm, null_selection_behavior="emit_null"), pc.take(d, didx), pc.take(d, didx32),
               pc.filter(d.slice(11, len(a) - 50), dm.slice(11, len(a) - 50)), pc.take(d.slice(7), didx32), pc.drop_null(d))
        for i, (g, w) in enumerate(zip(got, want[name])):
            assert g.type == w.type and not g.buffers()[1].is_cpu, (name, i, g.type)
            gh = to_host(g)
            assert gh.equals(w) and gh.null_count == w.null_count, (name, i, gh.slice(0, 5), w.slice(0, 5))
        # host arrays through the plugged registry: the reference kernels
        assert pc.filter(a, m).equals(want[name][0]) and pc.take(a, idx).equals(want[name][2])
        try:
            pc.take(d, to_device(pa.array([0, len(a)], pa.int64())))
            raise SystemExit("expected an index error")
        except pa.lib.ArrowIndexError as e:
            assert "out of bounds" in str(e), e
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1) - g0 >= 7 * len(cols) - 4
    print("LARGE_BINARY_OK")
""")


NESTED_SELECTION_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # VERDICT r3 missing 6: filter / take of fixed_size_list / list / large_list on device-resident arrays where the nested values
    # are fixed-width and free of nulls (FSLTakeExec -> FixedWidthTakeExec, vector_selection_internal.cc:991-1003;
    # ListSelectionImpl :620-760): embeddings, coordinate tuples, per-row number lists
    rng = np.random.default_rng(131)
    n = SC(200_000)
    def fsl(values, k, null_p=0.1):
        m = rng.random(len(values) // k) < null_p if null_p else None
        return pa.FixedSizeListArray.from_arrays(values, k, mask=None if m is None else pa.array(m))
    def lst(values, max_len, large=False, null_p=0.1):
        lens = rng.integers(0, max_len + 1, n)
        lens[rng.random(n) < 0.2] = 0
        offs = np.concatenate([[0], np.cumsum(lens)])
        vals = values(int(offs[-1]))
        m = pa.array(rng.random(n) < null_p) if null_p else None
        cls, odt = (pa.LargeListArray, np.int64) if large else (pa.ListArray, np.int32)
        return cls.from_arrays(pa.array(offs.astype(odt)), vals, mask=m)
    f32 = lambda k: pa.array(rng.standard_normal(k).astype(np.float32))
    cols = {
        "fsl_f32x4": fsl(f32(n * 4), 4),                                              # 16-byte rows
        "fsl_i16x3": fsl(pa.array(rng.integers(-9, 9, n * 3).astype(np.int16)), 3),   # 6-byte rows
        "fsl_f64x40": fsl(pa.array(rng.standard_normal((n // 8) * 40)), 40, 0.05),    # 320-byte rows (an embedding)
        "fsl_u8x1_dense": fsl(pa.array(rng.integers(0, 255, n).astype(np.uint8)), 1, 0),
        "fsl_of_fsl": fsl(fsl(pa.array(rng.integers(0, 255, n * 6).astype(np.uint8)), 3, 0), 2),   # nested: 6-byte rows
        "fsl_ts": fsl(pa.array(rng.integers(0, 10**12, n * 2), pa.timestamp("us")), 2),
        "list_i32": lst(lambda k: pa.array(rng.integers(-5, 5, k).astype(np.int32)), 7),
        "list_f64_dense": lst(lambda k: pa.array(rng.standard_normal(k)), 3, null_p=0),
        "large_list_u8": lst(lambda k: pa.array(rng.integers(0, 255, k).astype(np.uint8)), 20, large=True),
        "large_list_f32": lst(f32, 5, large=True),
    }
    mask = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.05)
    # (slices are re-based before the REFERENCE sees them: its fixed-width path mis-addresses a sliced fixed_size_list of
    #  fixed_size_lists — util/fixed_width_internal.cc:168-198 scales the outer offset by the list size twice; see
    #  plugin/selection_nested.inc.  The device arrays are sliced as they are.)
    def runs(a, mask, idx, idx32, rebase=lambda x: x):
        ln = len(a)
        m = mask.slice(0, ln)
        return (pc.filter(a, m), pc.filter(a, m, null_selection_behavior="emit_null"), pc.take(a, idx), pc.take(a, idx32),
                pc.filter(rebase(a.slice(11, ln - 50)), m.slice(11, ln - 50)), pc.take(rebase(a.slice(7)), idx32), pc.drop_null(a))
    idx_of = lambda ln: (pa.array(rng.integers(0, ln // 2, ln // 3), pa.int64(), mask=rng.random(ln // 3) < 0.1),
                         pa.array(rng.integers(0, ln // 2, 1000).astype(np.uint32)))
    idxs = {name: idx_of(len(a)) for name, a in cols.items()}
    want = {name: runs(a, mask, *idxs[name], rebase=lambda x: pa.concat_arrays([x, x.slice(0, 0)])) for name, a in cols.items()}
    nested = cols["fsl_of_fsl"]
    assert want["fsl_of_fsl"][5][0].as_py() == nested[7 + idxs["fsl_of_fsl"][1][0].as_py()].as_py()      # the re-based reference is right
    assert pc.take(nested.slice(7), pa.array([0]))[0].as_py() != nested[7].as_py(), "the reference's sliced nested fixed_size_list take got fixed"
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(x):
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    g0 = lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1)
    dmask = to_device(mask)
    for name, a in cols.items():
        d = to_device(a)
        got = runs(d, dmask, *(to_device(i) for i in idxs[name]))
        for i, (g, w) in enumerate(zip(got, want[name])):
            assert g.type == w.type, (name, i, g.type, w.type)
            gh = to_host(g)
            gh.validate(full=True)
            assert gh.equals(w) and gh.null_count == w.null_count, (name, i, gh.slice(0, 3), w.slice(0, 3))
        # host arrays through the plugged registry: the reference kernels
        assert pc.filter(a, mask.slice(0, len(a))).equals(want[name][0]) and pc.take(a, idxs[name][0]).equals(want[name][2])
        try:
            pc.take(d, to_device(pa.array([0, len(a)], pa.int64())))
            raise SystemExit("expected an index error")
        except pa.lib.ArrowIndexError as e:
            assert "out of bounds" in str(e), e
    assert lib.arrow_amd_plugin_calls(b"array_filter", 1) + lib.arrow_amd_plugin_calls(b"array_take", 1) - g0 >= 7 * len(cols) - 8
    # what the device kernels do not take is refused by name: nested values with nulls, boolean / var-width children
    small = pa.array([True, False, True])
    for bad in (pa.FixedSizeListArray.from_arrays(pa.array([1, None, 3, 4, 5, 6], pa.int32()), 2),
                pa.FixedSizeListArray.from_arrays(pa.array([True, False] * 3), 2),
                pa.array([[1, None], [], [3]], pa.list_(pa.int64())),
                pa.array([["a"], [], ["b", "c"]], pa.list_(pa.utf8())),
                pa.array([[[1]], [], [[2, 3]]], pa.list_(pa.list_(pa.int8())))):
        for fn in (lambda x: pc.filter(x, to_device(small)), lambda x: pc.take(x, to_device(pa.array([0, 2])))):
            try:
                fn(to_device(bad))
                raise SystemExit(f"expected NotImplemented for {bad.type}")
            except pa.ArrowNotImplementedError as e:
                assert "arrow_amd" in str(e), e
        assert pc.filter(bad, small).equals(bad.take(pa.array([0, 2])))      # the same arrays on the host: the reference
    # empty inputs
    e = to_device(cols["fsl_f32x4"].slice(0, 0))
    assert len(pc.filter(e, to_device(pa.array([], pa.bool_())))) == 0 and len(pc.take(to_device(cols["list_i32"]), to_device(pa.array([], pa.int32())))) == 0
    print("NESTED_SELECTION_OK")
""")


FLOAT_GROUPED_SUM_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # hash_sum / hash_mean of float32 / float64: GroupedReducingAggregator adds every row to its group's DOUBLE accumulator in row
    # order (hash_aggregate_numeric.cc:70-83,196-206,352-430) — a sum whose bits depend on the order, so the comparison is with the
    # reference on ONE thread (one state, batches in order); the values' magnitudes differ by 30 orders so that any other order of
    # additions shows.  Under the stock GroupByNode (host and device-resident values) and in aggregate_rocm.
    rng = np.random.default_rng(61)
    n = SC(400_000)
    k = pa.array(rng.integers(0, 300, n).astype(np.int32), mask=rng.random(n) < 0.01)
    kw = pa.array(rng.integers(0, max(n // 3, 2), n))                      # many small groups
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-15, 15, n)
    x[rng.random(n) < 0.03] = 0.0
    x[rng.random(n) < 0.03] = -0.0
    kk = np.asarray(k.fill_null(0))
    xm = rng.random(n) < 0.15
    xm[kk == 9] = True                                                      # a group of nulls only
    f64 = pa.array(x, mask=xm)
    f32 = pa.array((rng.standard_normal(n) * 10.0 ** rng.integers(-12, 12, n)).astype(np.float32), mask=rng.random(n) < 0.1)
    big = rng.standard_normal(n) * 1e300
    big[kk == 11] = 1e308                                                   # a group whose sum overflows to inf
    f64b = pa.array(big)
    t = pa.table({"k": k, "kw": kw, "f64": f64, "f32": f32, "big": f64b})
    tc = pa.concat_tables([t.slice(0, n // 3), t.slice(n // 3, n // 5), t.slice(n // 3 + n // 5)])      # several batches, in order
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    aggs = [(c, fn, o) for c in ("f64", "f32", "big") for fn in ("sum", "mean") for o in (None, strict)]

    def run(tab, key):
        return tab.group_by(key, use_threads=False).aggregate(aggs).sort_by(key)

    def same_bits(a, b, what):
        a, b = (pa.concat_arrays(z.chunks) if isinstance(z, pa.ChunkedArray) else z for z in (a, b))
        assert a.type == b.type == pa.float64(), (what, a.type, b.type)
        assert a.is_null().equals(b.is_null()), (what, "validity", a.null_count, b.null_count)
        x, y = (np.asarray(z.fill_null(0.0)).view(np.uint64) for z in (a, b))
        bad = np.nonzero(x != y)[0]
        assert len(bad) == 0, (what, len(bad), a.take(pa.array(bad[:4])), b.take(pa.array(bad[:4])))

    want = {(name, key): run(tab, key) for name, tab in (("t", t), ("tc", tc)) for key in ("k", "kw")}
    w0 = want[("t", "k")]
    first = lambda tab, name: tab.column(tab.schema.names.index(name))      # (the same name twice: default and strict options)
    assert np.isinf(first(w0, "big_sum")[w0.column("k").to_pylist().index(11)].as_py())
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    gpu0, stock0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1), lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    for (name, key), w in want.items():
        got = run(t if name == "t" else tc, key)
        assert got.schema.equals(w.schema), (got.schema, w.schema)
        assert got.column(key).equals(w.column(key))
        for ci, col in enumerate(w.schema.names):
            if col != key:
                same_bits(got.column(ci), w.column(ci), (name, key, col, ci))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - gpu0 >= 4 * len(aggs), "the float sum vtables did not run on the device"
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock0

    # ---- device-resident value columns under the stock GroupByNode, and the aggregate_rocm node (host and device tables)
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    th = t.combine_chunks()
    td_vals = pa.table({"k": th.column("k").chunk(0), **{c: to_device(th.column(c).chunk(0)) for c in ("f64", "f32", "big")}})
    td_all = pa.table({c: to_device(th.column(c).chunk(0)) for c in ("k", "f64", "f32", "big")})
    daggs = [(c, "hash_" + fn, o, "%s_%s" % (c, fn)) for c in ("f64", "f32", "big") for fn in ("sum", "mean") for o in (None,)]
    def plan(tab, node="aggregate"):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(daggs, keys=["k"]))]).to_table(use_threads=False).sort_by("k")
    wh = want[("t", "k")]
    for tab, node, what in ((td_vals, "aggregate", "device values, stock GroupByNode"), (th, "aggregate_rocm", "aggregate_rocm host"),
                            (td_all, "aggregate_rocm", "aggregate_rocm device")):
        g = plan(tab, node)
        assert g.column("k").equals(wh.column("k")), what
        for c in ("f64", "f32", "big"):
            for fn in ("sum", "mean"):
                same_bits(g.column("%s_%s" % (c, fn)), first(wh, "%s_%s" % (c, fn)), (what, c, fn))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock0, "a float sum reached a reference kernel"
    print("FLOAT_GROUPED_SUM_OK")
""")


COUNT_DISTINCT_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    # hash_count_distinct in aggregate_rocm: GroupedCountDistinctImpl (kernels/hash_aggregate.cc:1400-1478) is a Grouper over
    # (value, group id) pairs whose uniques are counted per group — the device Grouper does the same over the staged value column
    # and the rows' group ids; the three CountOptions modes; -0.0 / 0.0 and NaN payloads are distinct by their bytes, as there
    rng = np.random.default_rng(71)
    n = SC(300_000)
    f = (rng.integers(0, 9, n).astype(np.float64) / 4)
    f[rng.random(n) < 0.05] = -0.0
    f[rng.random(n) < 0.05] = np.nan
    t = pa.table({
        "k": pa.array(rng.integers(0, 300, n).astype(np.int32), mask=rng.random(n) < 0.02),
        "k2": pa.array(rng.integers(0, 3, n).astype(np.int8)),
        "s": pa.array(["key%d" % i for i in rng.integers(0, 50, n)], pa.utf8()),
        "i64": pa.array(rng.integers(0, 40, n), mask=rng.random(n) < 0.1),
        "wide": pa.array(rng.integers(-2**62, 2**62, n)),                  # almost every pair distinct
        "i8": pa.array(rng.integers(-3, 3, n).astype(np.int8), mask=rng.random(n) < 0.3),
        "f64": pa.array(f, mask=rng.random(n) < 0.05),
        "d32": pa.array(rng.integers(0, 5, n).astype(np.int32), pa.date32()),
        "ts": pa.array(rng.integers(0, 7, n) * 10**9, pa.timestamp("ns")),
    })
    vals = ["i64", "wide", "i8", "f64", "d32", "ts"]
    aggs = [(c, "hash_count_distinct", pc.CountOptions(mode=m), "%s_%s" % (c, m)) for c in vals for m in ("only_valid", "only_null", "all")]
    aggs += [("i64", "hash_sum", None, "sum"), ([], "hash_count_all", None, "rows")]
    def plan(tab, node, keys):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=False).sort_by([(k, "ascending") for k in keys])
    key_sets = (["k"], ["k2", "k"], ["s"])
    want = {tuple(ks): plan(t, "aggregate", ks) for ks in key_sets}
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 2 + 3), t.slice(n // 2 + 3))])
    g0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    for ks in key_sets:
        w = want[tuple(ks)]
        for tab, what in ((t, "host"), (td, "device")):
            g = plan(tab, "aggregate_rocm", ks)
            assert g.schema.equals(w.schema), (g.schema, w.schema)
            for ci, name in enumerate(w.schema.names):
                assert g.column(ci).equals(w.column(ci)), (what, ks, name, g.column(ci).slice(0, 6), w.column(ci).slice(0, 6))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - g0 >= 2 * len(key_sets)
    for bad, text in ((pa.table({"k": [1, 2], "v": ["a", "b"]}), "fixed-width"), (pa.table({"k": [1, 2], "v": pa.array([1, 2], pa.decimal128(20, 2))}), "fixed-width")):
        try:
            acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(bad)),
                acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_count_distinct", None, "c")], keys=["k"]))]).to_table()
            raise SystemExit("expected NotImplemented")
        except pa.ArrowNotImplementedError as e:
            assert text in str(e), e
    print("COUNT_DISTINCT_OK")
""")


DECIMAL_SUM_SCRIPT = textwrap.dedent(r"""
    import ctypes, decimal, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # (the emulated device is one fiber scheduler: one Acero thread)
        pa.set_cpu_count(1)
        pa.set_io_thread_count(1)
    # hash_sum of decimal128 columns: GroupedSumImpl<Decimal128Type> keeps a Decimal128 per group, adds modulo 2^128 and widens the
    # output to precision 38 (hash_aggregate_numeric.cc:44-215) — under the stock GroupByNode (host and device-resident values,
    # several batches, threads: the sum does not depend on the order) and in aggregate_rocm
    rng = np.random.default_rng(83)
    n = SC(300_000)
    def dec(lo, hi, scale, prec, null_p):
        ints = rng.integers(lo, hi, n)
        return pa.array([decimal.Decimal(int(x)).scaleb(-scale) for x in ints], pa.decimal128(prec, scale), mask=rng.random(n) < null_p)
    big = [decimal.Decimal(int(a) * 10**18 + int(b)).scaleb(-4) for a, b in zip(rng.integers(-10**15, 10**15, n), rng.integers(0, 10**18, n))]
    kk = rng.integers(0, 300, n)
    pmask = rng.random(n) < 0.15
    pmask[kk == 9] = True                                                 # a group of nulls only
    t = pa.table({
        "k": pa.array(kk.astype(np.int32), mask=rng.random(n) < 0.01),
        "kw": pa.array(rng.integers(0, max(n // 3, 2), n)),
        "price": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**11, 10**11, n)], pa.decimal128(15, 2), mask=pmask),
        "big": pa.array(big, pa.decimal128(38, 4)),                      # group sums beyond 64 bits
        "tiny": dec(-5, 5, 0, 3, 0.0),
    })
    tc = pa.concat_tables([t.slice(0, n // 3), t.slice(n // 3, n // 5), t.slice(n // 3 + n // 5)])
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    # (+ the mean: the same sums divided by the counts, truncating, then rounded half away from zero — GroupedMeanImpl::DoMean;
    #  the mean keeps the input's decimal type)
    aggs = [(c, fn, o) for c in ("price", "big", "tiny") for fn in ("sum", "mean") for o in (None, strict)]
    def run(tab, key, threads):
        return tab.group_by(key, use_threads=threads).aggregate(aggs).sort_by(key)
    want = {(name, key): run(tab, key, False) for name, tab in (("t", t), ("tc", tc)) for key in ("k", "kw")}
    assert want[("t", "k")].schema.field(1).type == pa.decimal128(38, 2)
    # min_count = 0 and a group without a valid value (VERDICT r5 weak 9): GroupedMeanImpl::Finish divides EVERY group whose
    # count reaches min_count, the reference's BasicDecimal128::Divide fails the whole Finalize — the stock kernels first
    zero = pc.ScalarAggregateOptions(skip_nulls=True, min_count=0)
    tz = pa.table({"k": pa.array([1, 1, 2], pa.int32()), "v": pa.array([None, None, decimal.Decimal("1.50")], pa.decimal128(10, 2))})
    try:
        tz.group_by("k", use_threads=False).aggregate([("v", "mean", zero)])
        raise AssertionError("the reference's decimal hash_mean of an empty group with min_count = 0 is expected to fail")
    except pa.ArrowInvalid as e:
        ref_zero_error = str(e)
        assert "Division by 0 in Decimal" in ref_zero_error, ref_zero_error
    ref_zero_sum = tz.group_by("k", use_threads=False).aggregate([("v", "sum", zero)]).sort_by("k").column("v_sum").to_pylist()
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    gpu0, stock0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1), lib.arrow_amd_plugin_calls(b"hash_sum", 0)
    for (name, key), w in want.items():
        for threads in (False, True):
            got = run(t if name == "t" else tc, key, threads)
            assert got.schema.equals(w.schema), (got.schema, w.schema)
            for ci in range(w.num_columns):
                assert got.column(ci).equals(w.column(ci)), (name, key, threads, w.schema.names[ci], got.column(ci).slice(0, 4), w.column(ci).slice(0, 4))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - gpu0 >= 8 * len(aggs), "the decimal sum vtable did not run on the device"
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock0
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    th = t.combine_chunks()
    td_vals = pa.table({"k": th.column("k").chunk(0), **{c: to_device(th.column(c).chunk(0)) for c in ("price", "big", "tiny")}})
    td_all = pa.table({c: to_device(th.column(c).chunk(0)) for c in ("k", "price", "big", "tiny")})
    daggs = [(c, "hash_" + fn, None, c + "_" + fn) for c in ("price", "big", "tiny") for fn in ("sum", "mean")]
    def plan(tab, node):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(daggs, keys=["k"]))]).to_table(use_threads=False).sort_by("k")
    wh = plan(th, "aggregate")
    for tab, node, what in ((td_vals, "aggregate", "device values, stock GroupByNode"), (th, "aggregate_rocm", "aggregate_rocm host"),
                            (td_all, "aggregate_rocm", "aggregate_rocm device")):
        g = plan(tab, node)
        assert g.schema.equals(wh.schema), (what, g.schema, wh.schema)
        for c in wh.schema.names:
            assert g.column(c).equals(wh.column(c)), (what, c, g.column(c).slice(0, 4), wh.column(c).slice(0, 4))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 0) == stock0, "a decimal sum reached a reference kernel"
    assert wh.schema.field("price_mean").type == pa.decimal128(15, 2) and wh.schema.field("price_sum").type == pa.decimal128(38, 2)
    assert any(x is not None and x != 0 and (x.as_tuple().digits[-1] % 2) for x in wh.column("tiny_mean").to_pylist())   # (means that needed rounding)
    # ... and the same failure / the same sums on every route the shim serves: the vtable under the stock GroupByNode (host and
    # device-resident values) and aggregate_rocm (host and device tables)
    tzd = pa.table({"k": tz.column("k").chunk(0), "v": to_device(tz.column("v").chunk(0))})
    tzd_all = pa.table({"k": to_device(tz.column("k").chunk(0)), "v": tzd.column("v").chunk(0)})
    def plan_z(tab, node, fn):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions([("v", fn, zero, "o")], keys=["k"]))]).to_table(use_threads=False).sort_by("k")
    for tab, node, what in ((tz, "aggregate", "host, stock node"), (tzd, "aggregate", "device values, stock node"),
                            (tz, "aggregate_rocm", "aggregate_rocm host"), (tzd_all, "aggregate_rocm", "aggregate_rocm device")):
        try:
            plan_z(tab, node, "hash_mean")
            raise AssertionError("decimal hash_mean of an empty group with min_count = 0 must fail as the reference's does: " + what)
        except pa.ArrowInvalid as e:
            assert "Division by 0 in Decimal" in str(e), (what, str(e), ref_zero_error)
        assert plan_z(tab, node, "hash_sum").column("o").to_pylist() == ref_zero_sum, what
    # hash_min / hash_max of decimal128 in aggregate_rocm (GroupedMinMaxImpl<Decimal128Type>: signed 128-bit order; the rows
    # sorted by group id, one owner per group), default and strict options, host and device tables, two key shapes
    mm = [(c, "hash_" + fn, o, "%s_%s_%d" % (c, fn, o is strict)) for c in ("price", "big", "tiny") for fn in ("min", "max") for o in (None, strict)]
    def plan_mm(tab, node, keys):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(mm, keys=keys))]).to_table(use_threads=False).sort_by([(k, "ascending") for k in keys])
    thw = pa.table({"kw": th.column("kw").chunk(0), **{c: th.column(c).chunk(0) for c in ("price", "big", "tiny")}})
    tdw = pa.table({"kw": to_device(th.column("kw").chunk(0)), **{c: td_all.column(c).chunk(0) for c in ("price", "big", "tiny")}})
    for keys, host_tab, dev_tab in ((["k"], th, td_all), (["kw"], thw, tdw)):
        wm = plan_mm(host_tab, "aggregate", keys)
        for tab, what in ((host_tab, "host"), (dev_tab, "device")):
            gm = plan_mm(tab, "aggregate_rocm", keys)
            assert gm.schema.equals(wm.schema), (what, gm.schema, wm.schema)
            for c in wm.schema.names:
                assert gm.column(c).equals(wm.column(c)), ("decimal extrema", what, keys, c, gm.column(c).slice(0, 4), wm.column(c).slice(0, 4))
    # ---- the scalar aggregates of decimal128 device columns (SumImpl / MeanImpl / MinMaxImpl<Decimal128Type>): sum widened to
    # precision 38, mean in the input's type (truncating division, rounded half away from zero; null for no value), min_max /
    # min / max; chunked device arrays (several batches, merged states), options, all-null and empty columns
    r0 = lib.arrow_amd_plugin_calls(b"reduce", 1)
    scalar_cols = {c: th.column(c).chunk(0) for c in ("price", "big", "tiny")}
    scalar_cols["nulls"] = pa.array([None] * 100, pa.decimal128(12, 3))
    scalar_cols["empty"] = pa.array([], pa.decimal128(7, 1))
    for c, host in scalar_cols.items():
        dev_arr = to_device(host)
        half = len(host) // 2
        dev_chunked = pa.chunked_array([to_device(host.slice(0, half)), to_device(host.slice(half))]) if len(host) > 1 else None
        for opt in (None, pc.ScalarAggregateOptions(skip_nulls=False, min_count=1), pc.ScalarAggregateOptions(skip_nulls=True, min_count=len(host) + 1),
                    pc.ScalarAggregateOptions(skip_nulls=True, min_count=0)):
            for fn in (pc.sum, pc.mean, pc.min_max, pc.min, pc.max):
                w = fn(host, options=opt)
                for d in (dev_arr, dev_chunked):
                    if d is None:
                        continue
                    g = fn(d, options=opt)
                    assert g.type == w.type and g.equals(w), (c, fn.__name__, opt, g, w)
    assert lib.arrow_amd_plugin_calls(b"reduce", 1) - r0 >= 5 * 4 * 3
    # ---- decimal128 SORT KEYS (VERDICT r4 missing 2): a device-resident decimal array sorts as the key pair (high word int64,
    # low word uint64) through the existing sort — array_sort_indices / sort_indices of arrays, both orders and null placements,
    # values beyond 64 bits, negative values, ties (stable), slices, all-null and empty arrays; and as an order_by_rocm key
    s0 = lib.arrow_amd_plugin_calls(b"array_sort_indices", 1)
    def to_host(x):
        if all(b is None or b.is_cpu for b in x.buffers()):
            return x
        c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
        x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        lib.arrow_amd_copy_to_host.argtypes = [ctypes.c_void_p] * 4
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
    sort_cols = dict(scalar_cols)
    sort_cols["ties"] = pa.array([decimal.Decimal(int(x)).scaleb(-1) for x in rng.integers(-4, 4, 5000)], pa.decimal128(9, 1), mask=rng.random(5000) < 0.2)
    sort_cols["sliced"] = th.column("big").chunk(0).slice(7, max(len(th) // 2, 1))
    for c, host in sort_cols.items():
        dev_arr = to_device(host)
        for order in ("ascending", "descending"):
            for placement in ("at_end", "at_start"):
                w = pc.array_sort_indices(host, order=order, null_placement=placement)
                g = to_host(pc.array_sort_indices(dev_arr, order=order, null_placement=placement))
                assert g.equals(w), (c, order, placement, g.slice(0, 8), w.slice(0, 8))
        w = pc.sort_indices(host, sort_keys=[("x", "descending")], null_placement="at_start") if False else pc.sort_indices(host)
        assert to_host(pc.sort_indices(dev_arr)).equals(w), c
    assert lib.arrow_amd_plugin_calls(b"array_sort_indices", 1) - s0 >= 2 * 4 * len(sort_cols)
    ob = lambda tab, node: acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
        acero.Declaration(node, acero.OrderByNodeOptions([("tiny", "descending"), ("big", "ascending")], null_placement="at_start"))]).to_table(use_threads=False)
    wo = ob(th.select(["tiny", "big", "price"]), "order_by")
    go = ob(td_all.select(["tiny", "big", "price"]), "order_by_rocm")
    go = pa.table({name: pa.chunked_array([to_host(ch) for ch in go.column(name).chunks], go.schema.field(name).type) for name in go.schema.names})
    assert go.equals(wo), (go.slice(0, 5), wo.slice(0, 5))
    print("DECIMAL_SUM_OK")
""")


WIDE_KEYS_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(77)
    n = SC(1_000_000)
    pool = rng.integers(-2**62, 2**62, 40)
    t = pa.table({
        "k64": pa.array(pool[rng.integers(0, 40, n)], mask=rng.random(n) < 0.01),
        "l64": pa.array(pool[rng.integers(0, 3, n)], mask=rng.random(n) < 0.2),
        "m64": pa.array(rng.integers(0, 4, n), mask=rng.random(n) < 0.05),
        "a": pa.array(rng.integers(-3, 3, n).astype(np.int32), mask=rng.random(n) < 0.02),
        "b": pa.array(rng.integers(0, 4, n).astype(np.int16)),
        "c": pa.array(rng.integers(0, 3, n).astype(np.uint8), mask=rng.random(n) < 0.1),
        "d": pa.array(rng.integers(0, 5, n).astype(np.int32), pa.date32()),
        "ts": pa.array(rng.integers(0, 3, n) * 86_400_000_000, pa.timestamp("us")),
        "f": pa.array(rng.integers(0, 3, n).astype(np.float64) / 4, mask=rng.random(n) < 0.05),
        **{f"u{i}": pa.array(rng.integers(0, 2, n).astype(np.uint8), mask=(rng.random(n) < 0.1) if i % 3 == 0 else None) for i in range(10)},
        "v": pa.array(rng.integers(-2**36, 2**36, n), mask=rng.random(n) < 0.15),
        "w": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.05),
    })
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    plans = [
        (["k64", "l64", "m64"], [("v", "hash_sum", None, "s"), ("v", "hash_count", None, "c"), ([], "hash_count_all", None, "all")]),          # 24 bytes: two tables
        (["k64", "a", "d", "b"], [("v", "hash_sum", None, "s"), ("w", "hash_max", strict, "mx")]),                                          # 18 bytes (the row round 2 refused)
        (["l64", "m64", "ts", "f", "a"], [("v", "hash_min", None, "mn"), ("v", "hash_mean", None, "me")]),                                   # 36 bytes: three tables
        ([f"u{i}" for i in range(10)], [("w", "hash_sum", None, "s")]),                                                                    # 10 bytes but 10 columns: 8 + (id, 2)
        (["c", "k64", "b", "l64", "a", "m64", "d"], [("v", "hash_sum", strict, "s"), ("c", "hash_count", pc.CountOptions(mode="only_null"), "cn")]),
    ]
    def run(tab, node, keys, aggs):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=False).sort_by([(k, "ascending") for k in keys])
    want = [run(t, "aggregate", keys, aggs) for keys, aggs in plans]      # the reference GroupByNode, before registration
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def same(got, w, tag):
        assert got.schema.equals(w.schema), (tag, got.schema, w.schema)
        assert got.num_rows == w.num_rows, (tag, got.num_rows, w.num_rows)
        for i in range(w.num_columns):
            assert got.column(i).equals(w.column(i)), (tag, w.schema.names[i], got.column(i).slice(0, 5), w.column(i).slice(0, 5))

    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 2 + 3), t.slice(n // 2 + 3))])
    g0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    for (keys, aggs), w in zip(plans, want):
        same(run(t, "aggregate_rocm", keys, aggs), w, ("host", keys))
        same(run(td, "aggregate_rocm", keys, aggs), w, ("device", keys))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - g0 >= 2 * len(plans), "aggregate_rocm did not run the device Grouper"
    # the order of the groups is the order of first appearance of the whole key row, whatever the number of tables
    keys = ["k64", "l64", "m64"]
    got = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([([], "hash_count_all", None, "all")], keys=keys))]).to_table(use_threads=False)
    rows = list(zip(*[t.column(k).to_pylist() for k in keys]))
    first = list(dict.fromkeys(rows))
    assert list(zip(*[got.column(k).to_pylist() for k in keys])) == first
    try:
        run(t, "aggregate_rocm", ["a"] * 33, [("v", "hash_sum", None, "s")])
        raise SystemExit("aggregate_rocm accepted 33 keys")
    except pa.ArrowNotImplementedError as e:
        assert "1 to 32 keys" in str(e), e
    print("WIDE_KEYS_OK")
''')

STRING_KEYS_SCRIPT = textwrap.dedent(r'''
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    rng = np.random.default_rng(78)
    n = SC(600_000)
    # strings that share long prefixes, differ only in their last byte / only in length, contain NUL bytes, are empty, are
    # null; lengths 0 .. 45 (four 12-byte chunks) in "s", short ones in "t", binary in "bn"
    words = ["", "a", "a\x00", "a\x00\x00", "ab", "abcdefghijkl", "abcdefghijklm", "abcdefghijkl\x00", "abcdefghijklmnopqrstuvwx",
             "abcdefghijklmnopqrstuvwy", "abcdefghijklmnopqrstuvwxyz0123456789ABCDEFGHI", "abcdefghijklmnopqrstuvwxyz0123456789ABCDEFGHJ",
             "\u00e9t\u00e9", "zz", "0123456789ab", "0123456789a"]
    pick = rng.integers(0, len(words), n)
    t = pa.table({
        "s": pa.array([words[i] for i in pick], pa.utf8(), mask=rng.random(n) < 0.05),
        "t": pa.array([("k%d" % i) for i in rng.integers(0, 30, n)], pa.utf8(), mask=rng.random(n) < 0.02),
        "bn": pa.array([bytes([i % 3, 0, i % 2]) * (i % 5) for i in rng.integers(0, 60, n)], pa.binary()),
        "a": pa.array(rng.integers(-3, 3, n).astype(np.int32), mask=rng.random(n) < 0.02),
        "k64": pa.array(rng.integers(0, 4, n) << 40, mask=rng.random(n) < 0.1),
        "v": pa.array(rng.integers(-2**36, 2**36, n), mask=rng.random(n) < 0.15),
    })
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=2)
    plans = [
        (["s"], [("v", "hash_sum", None, "s_"), ("v", "hash_count", None, "c"), ([], "hash_count_all", None, "all")]),
        (["t"], [("v", "hash_min", None, "mn"), ("v", "hash_max", strict, "mx")]),
        (["a", "s"], [("v", "hash_sum", None, "s_")]),
        (["s", "k64", "t"], [("v", "hash_sum", strict, "s_"), ("v", "hash_mean", None, "me")]),
        (["bn", "t", "a"], [("v", "hash_sum", None, "s_"), ("a", "hash_count", pc.CountOptions(mode="only_null"), "cn")]),
    ]
    def run(tab, node, keys, aggs, sort=True):
        out = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=False)
        return out.sort_by([(k, "ascending") for k in keys]) if sort else out
    want = [run(t, "aggregate", keys, aggs) for keys, aggs in plans]      # the reference GroupByNode, before registration
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def same(got, w, tag):
        assert got.schema.equals(w.schema), (tag, got.schema, w.schema)
        assert got.num_rows == w.num_rows, (tag, got.num_rows, w.num_rows)
        for i in range(w.num_columns):
            assert got.column(i).equals(w.column(i)), (tag, w.schema.names[i], got.column(i).slice(0, 5), w.column(i).slice(0, 5))

    chunks = pa.concat_tables([t.slice(0, n // 3), t.slice(n // 3, 7), t.slice(n // 3 + 7)])     # several host batches: Concatenate
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 2 + 3), t.slice(n // 2 + 3))])
    g0 = lib.arrow_amd_plugin_calls(b"hash_sum", 1)
    for (keys, aggs), w in zip(plans, want):
        same(run(chunks, "aggregate_rocm", keys, aggs), w, ("host", keys))
        same(run(td, "aggregate_rocm", keys, aggs), w, ("device", keys))
    assert lib.arrow_amd_plugin_calls(b"hash_sum", 1) - g0 >= 2 * len(plans), "aggregate_rocm did not run the device Grouper"
    # round 4: the strings entered the tables as (length, 64-bit hash) and the groups were verified against their first
    # rows' bytes; no batch above needed the exact chunk columns
    lib.arrow_amd_plugin_string_key_hash_collisions.restype = ctypes.c_int64
    assert lib.arrow_amd_plugin_string_key_hash_collisions() == 0
    # a hash of 3 bits: different strings of one length share it all the time -> the verification sees it and the batch is
    # grouped again by the exact chunk columns; 0 bits = the chunk columns from the start.  Same results either way.
    for bits in (3, 0):
        lib.arrow_amd_plugin_set_string_key_hash_bits(ctypes.c_int64(bits))
        c0 = lib.arrow_amd_plugin_string_key_hash_collisions()
        for (keys, aggs), w in zip(plans, want):
            same(run(td, "aggregate_rocm", keys, aggs), w, ("device", keys, "hash bits", bits))
        assert (lib.arrow_amd_plugin_string_key_hash_collisions() > c0) == (bits == 3), bits
    lib.arrow_amd_plugin_set_string_key_hash_bits(ctypes.c_int64(64))
    # long keys: 8 / 64 / 512 / 3000 bytes, many distinct values that share their first 500 bytes, odd start offsets
    m = SC(120_000)
    base = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    lens = rng.choice([8, 64, 512, 3000], m)
    ids = rng.integers(0, 5000, m)
    longs = [base[:l - 4] + int(i).to_bytes(4, "little") for l, i in zip(lens.tolist(), ids.tolist())]
    tl = pa.table({"b": pa.array(longs, pa.binary(), mask=rng.random(m) < 0.03), "v": pa.array(rng.integers(-2**40, 2**40, m))})
    wl = run(tl, "aggregate", ["b"], [("v", "hash_sum", None, "s_"), ([], "hash_count_all", None, "all")])
    tld = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(tl.column(j).chunk(0)) for j in range(2)], names=tl.schema.names)])
    same(run(tl, "aggregate_rocm", ["b"], [("v", "hash_sum", None, "s_"), ([], "hash_count_all", None, "all")]), wl, "long keys host")
    same(run(tld, "aggregate_rocm", ["b"], [("v", "hash_sum", None, "s_"), ([], "hash_count_all", None, "all")]), wl, "long keys device")
    # groups in order of first appearance, the unique strings byte for byte (NUL bytes, empty vs null)
    got = run(t, "aggregate_rocm", ["s", "a"], [([], "hash_count_all", None, "all")], sort=False)
    rows = list(zip(t.column("s").to_pylist(), t.column("a").to_pylist()))
    first = list(dict.fromkeys(rows))
    assert list(zip(got.column("s").to_pylist(), got.column("a").to_pylist())) == first
    import collections
    cnt = collections.Counter(rows)
    assert got.column("all").to_pylist() == [cnt[r] for r in first]
    # all-null and all-empty string keys, and an empty input
    z = pa.table({"s": pa.array([None, None, None], pa.utf8()), "e": pa.array(["", "", ""], pa.utf8()), "v": pa.array([1, 2, 3])})
    got = run(z, "aggregate_rocm", ["s", "e"], [("v", "hash_sum", None, "sum")], sort=False)
    assert got.to_pydict() == {"s": [None], "e": [""], "sum": [6]}, got.to_pydict()
    e = run(t.slice(0, 0), "aggregate_rocm", ["s", "a"], [("v", "hash_sum", None, "sum")])
    assert e.num_rows == 0 and e.schema.names == ["s", "a", "sum"], e.schema
    try:
        run(pa.table({"s": pa.array(["x"], pa.large_utf8()), "v": pa.array([1])}), "aggregate_rocm", ["s"], [("v", "hash_sum", None, "sum")])
        raise SystemExit("aggregate_rocm accepted large_utf8 keys")
    except pa.ArrowNotImplementedError as e:
        assert "utf8 / binary keys" in str(e), e
    print("STRING_KEYS_OK")
''')

ACERO_GUARD_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # (the emulated device is one fiber scheduler: one Acero thread)
        pa.set_cpu_count(1)
        pa.set_io_thread_count(1)
    # VERDICT r4 "Next round" 8: the UNMODIFIED plan (stock node names, Table.group_by) over a table whose KEY column lives in
    # HBM used to end in a segfault inside the reference's CPU Grouper (compute/row/grouper.cc:695).  With the plugin registered
    # it returns a table or a Status — never a signal.
    rng = np.random.default_rng(101)
    n = SC(200_000)
    t = pa.table({"k": pa.array(rng.integers(0, 500, n).astype(np.int32), mask=rng.random(n) < 0.02),
                  "k2": pa.array(rng.integers(-3, 3, n)),
                  "v": pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.1),
                  "s": pa.array(["x%d" % (i % 7) for i in range(n)])})
    def gb(tab, keys, aggs, threads=False):
        # what Table.group_by builds (table_source -> aggregate, the STOCK names), spelled out: pyarrow's Python wrapper refuses
        # device tables before Acero sees them, C++ and Declaration callers get no such check
        decl = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration("aggregate", acero.AggregateNodeOptions([(c, "hash_" + f, None, c + "_" + f) for c, f in aggs], keys=keys))])
        return decl.to_table(use_threads=threads).sort_by([(k, "ascending") for k in keys])
    A1 = [("v", "sum"), ("v", "count"), ("v", "min")]
    want = gb(t, ["k"], A1)
    want2 = gb(t, ["k2", "k"], [("v", "mean")])
    want_host_key = gb(t, ["s"], [("v", "sum")])
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_acero_guard.restype = ctypes.c_int64
    lib.arrow_amd_plugin_acero_guard.argtypes = [ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    assert lib.arrow_amd_plugin_acero_guard(0) == 1, "the default ExecFactoryRegistry of this build was not recognised"
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    td = pa.table({"k": to_device(t.column("k").chunk(0)), "k2": to_device(t.column("k2").chunk(0)),
                   "v": to_device(t.column("v").chunk(0)), "s": t.column("s").chunk(0)})
    took0 = lib.arrow_amd_plugin_acero_guard(1)
    # 1. table_source -> aggregate (stock names) with device-resident keys: served by aggregate_rocm
    for threads in (False, True):
        got = gb(td, ["k"], A1, threads)
        assert got.equals(want), (threads, got.slice(0, 4), want.slice(0, 4))
        got2 = gb(td, ["k2", "k"], [("v", "mean")], threads)
        assert got2.equals(want2), (threads, got2.slice(0, 4), want2.slice(0, 4))
    assert lib.arrow_amd_plugin_acero_guard(1) - took0 == 4, lib.arrow_amd_plugin_acero_guard(1) - took0
    # 2. the same through a filter + projection that keeps the key's name, and one that computes a new key from device columns
    def plan(tab, decls):
        return acero.Declaration.from_sequence([acero.Declaration("table_source", acero.TableSourceNodeOptions(tab))] + decls).to_table(use_threads=False)
    agg = acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["k"]))
    flt = acero.Declaration("filter", acero.FilterNodeOptions(pc.field("k2") >= 0))
    th, tdd = t.drop(["s"]), td.drop(["s"])       # (a host column cannot be filtered by a device mask: "mixed host / device")
    w = plan(th, [flt, agg]).sort_by("k")
    g = plan(tdd, [flt, agg]).sort_by("k")
    assert g.equals(w), (g.slice(0, 4), w.slice(0, 4))
    prj = acero.Declaration("project", acero.ProjectNodeOptions([pc.add(pc.field("k"), pc.field("k")), pc.field("v")], ["kk", "v"]))
    agg_kk = acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "hash_sum", None, "v_sum")], keys=["kk"]))
    w = plan(th, [prj, agg_kk]).sort_by("kk")
    g = plan(tdd, [prj, agg_kk]).sort_by("kk")
    assert g.equals(w), (g.slice(0, 4), w.slice(0, 4))
    # 3. host keys over device VALUE columns keep the reference's GroupByNode (the hash_* vtables serve the values)
    took1 = lib.arrow_amd_plugin_acero_guard(1)
    got = gb(td, ["s"], [("v", "sum")])
    assert got.equals(want_host_key), (got, want_host_key)
    assert lib.arrow_amd_plugin_acero_guard(1) == took1
    # 4. what aggregate_rocm does not serve is REFUSED by name (the stock node would have read the keys on the host)
    ref0 = lib.arrow_amd_plugin_acero_guard(2)
    for fn in ("hash_approximate_median", "hash_tdigest"):
        try:
            plan(td, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", fn, None, "x")], keys=["k"]))])
            raise SystemExit("a keyed %s over device-resident keys was not refused" % fn)
        except pa.ArrowNotImplementedError as e:
            assert "device-resident" in str(e), e
    assert lib.arrow_amd_plugin_acero_guard(2) - ref0 == 2
    # 4b. host keys (or none) over device VALUE columns: an aggregate whose kernel is the reference's own would read HBM through
    # host pointers inside the stock node.  Served by aggregate_rocm where it can (hash_list: a takeover), refused by name
    # where nothing serves it (hash_tdigest under a host key; the scalar variance / tdigest / first of a device column)
    took2, ref1 = lib.arrow_amd_plugin_acero_guard(1), lib.arrow_amd_plugin_acero_guard(2)
    got = plan(td, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "hash_list", None, "l")], keys=["s"]))]).sort_by("s")
    wnt = plan(t, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "hash_list", None, "l")], keys=["s"]))]).sort_by("s")
    assert got.equals(wnt)
    assert lib.arrow_amd_plugin_acero_guard(1) == took2 + 1
    for fn, keys in (("hash_tdigest", ["s"]), ("variance", None), ("tdigest", None), ("first", None)):
        try:
            plan(td, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", fn, None, "x")], keys=keys))])
            raise SystemExit("%s over device-resident values was not refused" % fn)
        except pa.ArrowNotImplementedError as e:
            assert "device-resident values" in str(e) and fn in str(e), e
    assert lib.arrow_amd_plugin_acero_guard(2) - ref1 == 4
    assert plan(t, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "variance", None, "x")]))]).num_rows == 1      # (host tables: the reference's)
    # 5. host tables are untouched, and so is a key-less aggregation over the device table
    assert gb(t, ["k"], A1).equals(want)
    assert t.group_by("k", use_threads=False).aggregate(A1).sort_by("k").equals(want.rename_columns(["k", "v_sum", "v_count", "v_min"]))
    assert plan(td, [acero.Declaration("aggregate", acero.AggregateNodeOptions([("v", "sum", None, "s")]))]).column("s")[0].as_py() == pc.sum(t.column("v")).as_py()
    print("ACERO_GUARD_OK")
""")


FIRST_LAST_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # (the emulated device is one fiber scheduler: one Acero thread)
        pa.set_cpu_count(1)
        pa.set_io_thread_count(1)
    # hash_first / hash_last / hash_one in aggregate_rocm: the first / last NON-NULL value of a group in row order
    # (GroupedFirstLastImpl kernels/hash_aggregate.cc:738-925, GroupedOneImpl :1556-1625), skip_nulls on and off, value types of
    # 1 to 16 bytes, groups of nulls only, nulls before / after the values — equal to the reference's GroupByNode (which runs
    # these ordered aggregates on one thread only), host and device-resident tables, several batches, int32 and utf8 keys
    rng = np.random.default_rng(131)
    n = SC(300_000)
    kk = rng.integers(0, 700, n)
    vmask = rng.random(n) < 0.3
    vmask[kk == 5] = True                                    # a group of nulls only
    import decimal
    t = pa.table({
        "k": pa.array(kk.astype(np.int32), mask=rng.random(n) < 0.01),
        "s": pa.array(["key-%d" % (x % 50) for x in kk]),
        "i64": pa.array(rng.integers(-2**60, 2**60, n), mask=vmask),
        "i8": pa.array(rng.integers(-100, 100, n).astype(np.int8), mask=rng.random(n) < 0.5),
        "u16": pa.array(rng.integers(0, 60000, n).astype(np.uint16)),
        "f32": pa.array(rng.random(n).astype(np.float32), mask=rng.random(n) < 0.2),
        "f64": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.2),
        "ts": pa.array(rng.integers(0, 10**15, n), pa.timestamp("us"), mask=rng.random(n) < 0.1),
        "d32": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32()),
        "small": pa.array(rng.integers(-3, 4, n).astype(np.int32), mask=rng.random(n) < 0.1),   # (products that stay small, with zeros)
        "dec": pa.array([decimal.Decimal(int(x)).scaleb(-3) for x in rng.integers(-10**12, 10**12, n)], pa.decimal128(20, 3), mask=rng.random(n) < 0.25),
    })
    tc = pa.concat_tables([t.slice(0, n // 7), t.slice(n // 7, n // 2), t.slice(n // 7 + n // 2)])
    vals = ["i64", "i8", "u16", "f32", "f64", "ts", "d32"]      # (the reference's hash_first_last has no decimal kernel; hash_one has)
    keep = pc.ScalarAggregateOptions(skip_nulls=False)
    aggs = [(c, "hash_" + f, o, "%s_%s_%d" % (c, f, o is keep)) for c in vals for f in ("first", "last") for o in (None, keep)]
    aggs += [(c, "hash_one", None, c + "_one") for c in vals + ["dec"]] + [("i64", "hash_sum", None, "sum"), ([], "hash_count_all", None, "rows")]
    # hash_product (GroupedProductImpl kernels/hash_aggregate_numeric.cc:311-347): wrapping int64 / uint64 products, double
    # products in ROW order (float32 values widened first), min_count / skip_nulls as the sums
    strict = pc.ScalarAggregateOptions(skip_nulls=False, min_count=3)
    aggs += [(c, "hash_product", o, "%s_prod_%d" % (c, o is strict)) for c in ("i64", "i8", "u16", "f32", "f64", "small") for o in (None, strict)]
    # hash_first_last (a struct of the two), hash_list (every value of the group in row order, nulls included), hash_distinct
    # (the distinct values in order of first appearance; CountOptions: without the null / only the null / with it)
    aggs += [(c, "hash_first_last", o, "%s_fl_%d" % (c, o is keep)) for c in ("i64", "f32") for o in (None, keep)]
    aggs += [(c, "hash_list", None, c + "_list") for c in ("i64", "i8", "f64", "ts")]
    # hash_min_max: struct<min, max> (the hash_min and hash_max of the column joined), integer / float / temporal / decimal values
    aggs += [(c, "hash_min_max", o, "%s_mm_%d" % (c, o is keep)) for c in ("i64", "i8", "u16", "f32", "ts", "dec") for o in (None, keep)]
    aggs += [(c, "hash_distinct", pc.CountOptions(mode=m), "%s_distinct_%s" % (c, m)) for c in ("small", "i8", "d32", "u16") for m in ("only_valid", "only_null", "all")]
    def plan(tab, node, keys, threads=False):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=threads).sort_by([(k, "ascending") for k in keys])
    key_sets = (["k"], ["s"], ["s", "k"])
    want = {tuple(ks): plan(t, "aggregate", ks) for ks in key_sets}
    w = want[("k",)]
    assert w.column("i64_first_0").null_count >= 1 and w.column("i64_first_1").null_count > w.column("i64_first_0").null_count   # (nulls first in some groups)
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 3 + 1), t.slice(n // 3 + 1))])
    threaded = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") != "1"
    for ks in key_sets:
        w = want[tuple(ks)]
        for tab, what, threads in ((t, "host", False), (tc, "host chunks", False), (td, "device", False)) + (((tc, "host chunks, threads", True),) if threaded else ()):
            if os.environ.get("ARROW_AMD_TEST_LIGHT") == "1" and what == "host":      # (the emulated tier: chunks and device tables)
                continue
            g = plan(tab, "aggregate_rocm", ks, threads)
            assert g.schema.equals(w.schema), (g.schema, w.schema)
            for ci, name in enumerate(w.schema.names):
                if "_distinct_" in name:
                    # "Order of sub-arrays is not stable" (acero/hash_aggregate_test.cc:2487: the reference's own test sorts every
                    # list first — its order is the swiss table's insertion order, ours the order of first appearance)
                    canon = lambda col: [None if x is None else sorted(x, key=lambda v: (v is None, 0 if v is None else v)) for x in col.to_pylist()]
                    assert canon(g.column(ci)) == canon(w.column(ci)) and g.column(ci).type == w.column(ci).type, (what, ks, name)
                    continue
                assert g.column(ci).equals(w.column(ci)), (what, ks, name, g.column(ci).slice(0, 8), w.column(ci).slice(0, 8))
    for bad in ("s",):
        try:
            acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
                acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([(bad, "hash_first", None, "x")], keys=["k"]))]).to_table()
            raise SystemExit("expected NotImplemented")
        except pa.ArrowNotImplementedError as e:
            assert "fixed-width" in str(e), e
    print("FIRST_LAST_OK")
""")


MOMENTS_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    from pyarrow import acero
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    path = build_plugin()
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # (the emulated device is one fiber scheduler: one Acero thread)
        pa.set_cpu_count(1)
        pa.set_io_thread_count(1)
    # hash_variance / hash_stddev / hash_skew / hash_kurtosis in aggregate_rocm (GroupedStatisticImpl,
    # kernels/hash_aggregate_numeric.cc:457-843): two passes over all rows of the node against the reference's per-batch
    # moments merged batch by batch — equal up to floating-point rounding, which is also how the reference's own tests compare
    # (acero/hash_aggregate_test.cc VarianceAndStddev / SkewAndKurtosis: AssertDatumsApproxEqual).  Where a group is null is exact:
    # count <= ddof, the unbiased skew / kurtosis of 2 / 3 values, min_count, a null seen with skip_nulls = false.
    rng = np.random.default_rng(151)
    n = SC(300_000)
    kk = rng.integers(0, 500, n)
    kk[:40] = np.arange(1000, 1040) // 4 * 4 + np.array([0, 0, 0, 0] * 10)      # groups of exactly four rows ...
    kk[40:46] = [2000, 2000, 2000, 2001, 2001, 2002]                               # ... of three, two, one
    vmask = rng.random(n) < 0.2
    vmask[kk == 7] = True                                                          # a group of nulls only
    vmask[:46] = False
    t = pa.table({
        "k": pa.array(kk.astype(np.int32)),
        "s": pa.array(["key-%d" % (x % 40) for x in kk]),
        "f64": pa.array(rng.standard_normal(n) * 1e3 + 1e6, mask=vmask),          # (a mean far from zero: the two passes matter)
        "f32": pa.array(rng.random(n).astype(np.float32), mask=rng.random(n) < 0.1),
        "i64": pa.array(rng.integers(-10**9, 10**9, n), mask=rng.random(n) < 0.1),
        "i16": pa.array(rng.integers(-3000, 3000, n).astype(np.int16)),
        "u8": pa.array(rng.integers(0, 256, n).astype(np.uint8), mask=rng.random(n) < 0.3),
        "u64": pa.array(rng.integers(0, 2**40, n).astype(np.uint64)),
    })
    tc = pa.concat_tables([t.slice(0, n // 7), t.slice(n // 7, n // 2), t.slice(n // 7 + n // 2)])
    # (the emulated CPU tier — ARROW_AMD_TEST_LIGHT — runs two of the six value types and two of the three tables: every
    #  aggregate is a handful of launches of thousands of emulated workgroups whatever the row count)
    light = os.environ.get("ARROW_AMD_TEST_LIGHT") == "1"
    vals = ["f64", "u8"] if light else ["f64", "f32", "i64", "i16", "u8", "u64"]
    V, S = pc.VarianceOptions, pc.SkewOptions
    var_opts = [None, V(ddof=1), V(ddof=3, min_count=5), V(ddof=0, skip_nulls=False), V(ddof=1, skip_nulls=False, min_count=2)]
    skew_opts = [None, S(biased=False), S(skip_nulls=False, biased=True, min_count=4), S(skip_nulls=True, biased=False, min_count=6)]
    aggs = [(c, "hash_" + f, o, "%s_%s_%d" % (c, f, i)) for c in vals for f in ("variance", "stddev") for i, o in enumerate(var_opts)]
    aggs += [(c, "hash_" + f, o, "%s_%s_%d" % (c, f, i)) for c in vals for f in ("skew", "kurtosis") for i, o in enumerate(skew_opts)]
    aggs += [("i64", "hash_sum", None, "sum"), ([], "hash_count_all", None, "rows"), ("f64", "hash_mean", None, "mean")]
    def plan(tab, node, keys, threads=False):
        return acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(tab)),
            acero.Declaration(node, acero.AggregateNodeOptions(aggs, keys=keys))]).to_table(use_threads=threads).sort_by([(k, "ascending") for k in keys])
    key_sets = (["k"], ["s", "k"])
    want = {tuple(ks): plan(t, "aggregate", ks) for ks in key_sets}
    w = want[("k",)]
    assert w.column("f64_variance_1").null_count >= 2 and w.column("f64_skew_1").null_count > w.column("f64_skew_0").null_count
    lib = ctypes.CDLL(path)
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(c.combine_chunks().column(j).chunk(0)) for j in range(t.num_columns)],
                                                           names=t.schema.names)
                                for c in (t.slice(0, n // 3 + 1), t.slice(n // 3 + 1))])
    threaded = os.environ.get("ARROW_AMD_PLUGIN_EMULATED") != "1"
    worst = 0.0
    for ks in key_sets:
        w = want[tuple(ks)]
        for tab, what, threads in ((t, "host", False), (tc, "host chunks", False), (td, "device", False)) + (((tc, "host chunks, threads", True),) if threaded else ()):
            if os.environ.get("ARROW_AMD_TEST_LIGHT") == "1" and what == "host":      # (the emulated tier: chunks and device tables)
                continue
            g = plan(tab, "aggregate_rocm", ks, threads)
            assert g.schema.equals(w.schema), (g.schema, w.schema)
            for ci, name in enumerate(w.schema.names):
                gc, wc = g.column(ci).combine_chunks(), w.column(ci).combine_chunks()
                if not any(f in name for f in ("_variance_", "_stddev_", "_skew_", "_kurtosis_")):
                    assert gc.equals(wc), (what, ks, name)
                    continue
                assert gc.is_valid().equals(wc.is_valid()), (what, ks, name, "where the statistic is null")
                a, b = (np.asarray(x.fill_null(0.0)) for x in (gc, wc))
                both_nan = np.isnan(a) & np.isnan(b)                       # (0 / 0 of a constant group, in the reference as here)
                # the tolerance: the moments of <= n values of ~1e-16 relative rounding each; m3 / m4 of nearly symmetric groups
                # are small differences of large terms, so skew / kurtosis get an absolute part
                tol = 1e-9 * np.maximum(np.abs(b), 1.0) if ("_skew_" in name or "_kurtosis_" in name) else 1e-11 * np.abs(b)
                bad = ~both_nan & ~(np.abs(a - b) <= tol)
                assert not bad.any(), (what, ks, name, a[bad][:4], b[bad][:4])
                ok = ~both_nan & (b != 0)
                if ok.any():
                    worst = max(worst, float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if "_variance_" in name or "_stddev_" in name else worst)
    assert worst < 1e-11, worst
    for fn, opts, msg in (("hash_variance", pc.ScalarAggregateOptions(), "VarianceOptions"), ("hash_skew", V(ddof=1), "SkewOptions")):
        try:
            acero.Declaration.from_sequence([
                acero.Declaration("table_source", acero.TableSourceNodeOptions(t)),
                acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("f64", fn, opts, "x")], keys=["k"]))]).to_table()
            raise SystemExit("expected TypeError")
        except pa.ArrowTypeError as e:
            assert msg in str(e), e
    print("MOMENTS_OK", "largest relative difference of a variance / stddev: %.2e" % worst)
""")


# (id, script, marker the script prints last, scale of the emulated run, what the case pins)
RANK_SELECT_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler, warnings
    faulthandler.enable()
    warnings.simplefilter("ignore", FutureWarning)        # (RankOptions' option-wide null_placement: deprecated in 25.0, same result)
    import numpy as np
    import pyarrow as pa, pyarrow.compute as pc
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    light = os.environ.get("ARROW_AMD_TEST_LIGHT") == "1"
    rng = np.random.default_rng(11)
    n = SC(400_000)
    fl = np.round(rng.standard_normal(n) * 4) / 4
    fl[rng.random(n) < 0.05] = np.nan
    fl[rng.random(n) < 0.02] = -0.0
    cols = {
        "i64": pa.array(rng.integers(-50, 50, n), mask=rng.random(n) < 0.1),
        "u32": pa.array(rng.integers(0, 2**32 - 1, n).astype(np.uint32)),
        "f64": pa.array(fl, mask=rng.random(n) < 0.1),
        "f32": pa.array(fl.astype(np.float32)),
        "ts": pa.array(rng.integers(0, 1000, n), pa.timestamp("ms"), mask=rng.random(n) < 0.05),
        "d32": pa.array(rng.integers(0, 300, n).astype(np.int32), pa.date32()),
    }
    combos = [(o, p) for o in ("ascending", "descending") for p in ("at_end", "at_start")]
    # ---- the reference: stock kernels, before anything is registered
    want_rank, want_q, want_n = {}, {}, {}
    for name, a in cols.items():
        for o, p in combos:
            for tb in ("min", "max", "first", "dense"):
                want_rank[name, o, p, tb] = pc.rank(a, sort_keys=o, null_placement=p, tiebreaker=tb)
            want_q[name, o, p] = pc.rank_quantile(a, sort_keys=o, null_placement=p)
            want_n[name, o, p] = pc.rank_normal(a, sort_keys=o, null_placement=p)
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_plugin_calls.restype = ctypes.c_int64
    lib.arrow_amd_plugin_calls.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(m) for m in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))
    dev = {name: to_device(a) for name, a in cols.items()}
    # pyarrow.compute's generated wrappers (pc.rank, pc.select_k_unstable, ...) hold the Function objects they found when the
    # module was imported; a function REPLACED in the registry afterwards (these MetaFunctions cannot be extended in place,
    # unlike the kernels of a VectorFunction) is reached by name: CallFunction in C++, pc.call_function here
    def rank(x, o, p, tb): return pc.call_function("rank", [x], pc.RankOptions(sort_keys=o, null_placement=p, tiebreaker=tb))
    def rank_quantile(x, o, p): return pc.call_function("rank_quantile", [x], pc.RankQuantileOptions(sort_keys=o, null_placement=p))
    def rank_normal(x, o, p): return pc.call_function("rank_normal", [x], pc.RankQuantileOptions(sort_keys=o, null_placement=p))
    def select_k(x, k, keys): return pc.call_function("select_k_unstable", [x], pc.SelectKOptions(k, keys))
    def partition_nth(x, pivot, p="at_end"): return pc.call_function("partition_nth_indices", [x], pc.PartitionNthOptions(pivot, null_placement=p))
    picked = combos[:1] + combos[3:] if light else combos
    # ---- rank / rank_quantile of device-resident arrays: ranks stay in HBM, equal to the reference's bit for bit
    g0 = lib.arrow_amd_plugin_calls(b"array_sort_indices", 1)
    for name in cols:
        for o, p in picked:
            for tb in ("min", "max", "first", "dense"):
                got = rank(dev[name], o, p, tb)
                assert not got.is_cpu and got.type == pa.uint64(), (name, got.type)
                assert to_host(got).equals(want_rank[name, o, p, tb]), (name, o, p, tb)
            gq = rank_quantile(dev[name], o, p)
            assert not gq.is_cpu and gq.type == pa.float64()
            a, b = (np.asarray(x).view(np.uint64) for x in (to_host(gq), want_q[name, o, p]))
            assert np.array_equal(a, b), (name, o, p, "rank_quantile")
            # rank_normal: NormalPPF of the quantile rank — within the reference's own bar for it (4 ULPs, util/math_test.cc)
            gn = rank_normal(dev[name], o, p)
            assert not gn.is_cpu and gn.type == pa.float64()
            from tests.parity_cases import max_ulps
            hn = to_host(gn)
            assert hn.null_count == 0 and max_ulps(np.asarray(hn), np.asarray(want_n[name, o, p])) <= 4, (name, o, p, "rank_normal")
    assert lib.arrow_amd_plugin_calls(b"array_sort_indices", 1) - g0 >= len(cols) * len(picked) * 6
    assert rank_normal(cols["f64"], "ascending", "at_end").equals(want_n["f64", "ascending", "at_end"])      # host: stock
    # chunked device input: ranks number the rows of the logical column
    cut = n // 3 + 1
    chunked = pa.chunked_array([to_device(cols["i64"].slice(0, cut)), to_device(cols["i64"].slice(cut))])
    got = rank(chunked, "descending", "at_end", "dense")
    assert to_host(got).equals(want_rank["i64", "descending", "at_end", "dense"])
    # host input: the stock function, untouched
    assert rank(cols["f64"], "ascending", "at_end", "min").equals(want_rank["f64", "ascending", "at_end", "min"])
    # empty, and a type without a device sort key: a Status, not a crash
    assert len(rank(to_device(pa.array([], pa.int64())), "ascending", "at_end", "first")) == 0
    try:
        rank(to_device(pa.array([True, False, None])), "ascending", "at_end", "first")
        raise SystemExit("rank of device booleans should be refused")
    except pa.ArrowNotImplementedError as e:
        assert "rank of device-resident" in str(e), e
    # ---- select_k_unstable: the VALUES at the selected rows are the head of the sorted order (ties may pick other rows)
    def values_at(a, idx):
        return a.take(idx)
    def same_values(x, y):
        assert x.is_null().equals(y.is_null())
        if pa.types.is_floating(x.type):
            xa, ya = (np.nan_to_num(np.asarray(pc.fill_null(z, 0.0)), nan=1e300) for z in (x, y))
            assert np.array_equal(xa, ya)
        else:
            assert x.equals(y)
    for name in ("i64", "f64", "ts"):
        for o, p in picked:
            for k in (0, 7, n // 5, n + 3):
                got = select_k(dev[name], k, [("x", o, p)] if p != "at_end" else [("x", o)])
                assert not got.is_cpu or len(got) == 0
                idx = to_host(got) if len(got) else pa.array([], pa.uint64())
                assert len(idx) == min(k, n)
                want = pc.array_sort_indices(cols[name], order=o, null_placement=p).slice(0, min(k, n))
                same_values(values_at(cols[name], idx), values_at(cols[name], want))
    # select_k by a THRESHOLD (round 6): no sort of the column — histogram, one comparison, the candidates' sort; index for
    # index the head of the stable sort (forced at this size; a key with few distinct values takes the sort instead)
    lib.arrow_amd_plugin_set_select_k_min_rows.argtypes = [ctypes.c_int64]
    lib.arrow_amd_plugin_select_k_threshold_runs.restype = ctypes.c_int64
    lib.arrow_amd_plugin_set_select_k_min_rows(0)
    wide = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
    dwide = to_device(wide)
    for arr, darr, name in ((wide, dwide, "wide"), (cols["ts"], dev["ts"], "ts")):
        for o in ("ascending", "descending"):
            for k in (1, 50, n // 20):
                t0 = lib.arrow_amd_plugin_select_k_threshold_runs()
                got = to_host(select_k(darr, k, [("x", o)]))
                want = pc.array_sort_indices(arr, order=o).slice(0, k)
                assert got.equals(want), (name, o, k)
                assert lib.arrow_amd_plugin_select_k_threshold_runs() - t0 == 1, (name, o, k)
    t0 = lib.arrow_amd_plugin_select_k_threshold_runs()
    few = pa.array(rng.integers(0, 3, n))
    assert to_host(select_k(to_device(few), 5, [("x", "ascending")])).equals(pc.array_sort_indices(few).slice(0, 5))
    assert lib.arrow_amd_plugin_select_k_threshold_runs() == t0
    lib.arrow_amd_plugin_set_select_k_min_rows(1 << 22)
    # a device-resident table with two keys
    tdev = pa.table({"a": dev["i64"], "b": dev["u32"]})
    thost = pa.table({"a": cols["i64"], "b": cols["u32"]})
    keys = [("a", "descending"), ("b", "ascending")]
    got = to_host(select_k(tdev, 1000, keys))
    want = pc.sort_indices(thost, sort_keys=keys).slice(0, 1000)
    assert got.equals(want)                                         # (b is unique with overwhelming odds: one valid answer)
    try:
        select_k(dev["i64"], -1, [("x", "ascending")])
        raise SystemExit("negative k should be refused")
    except pa.ArrowInvalid as e:
        assert "nonnegative" in str(e), e
    # ---- partition_nth_indices: a permutation with the pivot-th smallest in place, null-likes at their end
    for name in ("i64", "f64"):
        a = cols[name]
        vals = np.asarray(pc.fill_null(a, 0)).astype(np.float64)
        null_like = np.asarray(a.is_null()) | np.isnan(vals)
        cnt = int(null_like.sum())
        for p in ("at_end", "at_start"):
            for pivot in (0, n // 2, n):
                got = np.asarray(to_host(partition_nth(dev[name], pivot, p))).astype(np.int64)
                assert np.array_equal(np.sort(got), np.arange(n))
                nl = null_like[got]
                body, at = (got[:n - cnt], pivot) if p == "at_end" else (got[cnt:], pivot - cnt)
                assert (not nl[:n - cnt].any() and nl[n - cnt:].all()) if p == "at_end" else (nl[:cnt].all() and not nl[cnt:].any())
                if 0 <= at < len(body):
                    assert (vals[body[:at]] <= vals[body[at]]).all() and (vals[body[at:]] >= vals[body[at]]).all()
    # ---- boolean sort keys (round 6): the reference's counting sort, on the device — exact indices (the sort is stable)
    bools = pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.15)
    for arr in (bools, bools.slice(5, n // 2 + 3), pa.array(rng.random(n) < 0.5), pa.array([None, None], pa.bool_()), pa.array([], pa.bool_())):
        darr = to_device(arr) if len(arr) else arr
        if len(arr) and arr.offset:
            darr = to_device(bools).slice(arr.offset, len(arr))
        for o, p in combos:
            got = pc.array_sort_indices(darr, order=o, null_placement=p)
            want = pc.array_sort_indices(arr, order=o, null_placement=p)
            assert (to_host(got) if len(arr) else got).equals(want), (o, p, len(arr))
    # ---- dictionary sort keys (round 6): the reference's algorithm — ranks of the dictionary values, Take, sort of the ranks
    for name in ("i64", "f64"):
        dd = pc.dictionary_encode(dev[name])                 # a dictionary array whose indices AND dictionary live in HBM
        hd = pc.dictionary_encode(cols[name])
        assert pa.types.is_dictionary(dd.type) and not dd.is_cpu
        for o, p in combos:
            got = pc.array_sort_indices(dd, order=o, null_placement=p)
            assert not got.is_cpu and to_host(got).equals(pc.array_sort_indices(hd, order=o, null_placement=p)), (name, o, p)
    dd = pc.dictionary_encode(dev["i64"])
    sl = dd.slice(7, n // 2)
    assert to_host(pc.array_sort_indices(sl, order="descending")).equals(pc.array_sort_indices(pc.dictionary_encode(cols["i64"]).slice(7, n // 2), order="descending"))
    # ---- struct sort keys (round 6): the null structs to their end in row order, the others by their fields in turn
    # (ArrayCompareSorter<StructType> -> SortStructArray); fields with nulls of their own, a sliced struct, no null struct
    st_type = pa.struct([("a", pa.int64()), ("b", pa.float64()), ("s", pa.utf8()), ("h", pa.int16())])
    m = SC(40_000)
    fields_h = [pa.array(rng.integers(0, 6, m), mask=rng.random(m) < 0.1),
                pa.array(np.round(rng.standard_normal(m), 1), mask=rng.random(m) < 0.1),
                pa.array(np.array(["", "a", "ab", "b"], dtype=object)[rng.integers(0, 4, m)], pa.utf8(), mask=rng.random(m) < 0.1),
                pa.array(rng.integers(-3, 3, m).astype(np.int16))]
    fields_d = [to_device(f) for f in fields_h]
    for null_p in (0.15, 0.0):
        valid = pa.array(rng.random(m) >= null_p)
        host_struct = pa.StructArray.from_arrays(fields_h, fields=list(st_type), mask=pc.invert(valid) if null_p else None)
        bufs = [to_device(valid).buffers()[1]] if null_p else [None]
        dev_struct = pa.Array.from_buffers(st_type, m, bufs, null_count=host_struct.null_count, children=fields_d)
        for hs, ds in ((host_struct, dev_struct), (host_struct.slice(9, m // 2), dev_struct.slice(9, m // 2))):
            for o, p in (combos[:1] + combos[3:] if light else combos):
                got = pc.array_sort_indices(ds, order=o, null_placement=p)
                want = pc.array_sort_indices(hs, order=o, null_placement=p)
                assert not got.is_cpu and to_host(got).equals(want), ("struct", null_p, hs.offset, o, p)
    # ---- large_utf8 / large_binary sort keys (round 6): offsets narrowed on the device, then the utf8 chain
    words = [None if i % 13 == 0 else "w%05d" % int(x) if i % 3 else "w%d" % int(x) for i, x in enumerate(rng.integers(0, 3000, SC(60_000)))]
    for typ in (pa.large_utf8(), pa.large_binary()):
        ls = pa.array(words if typ == pa.large_utf8() else [None if w is None else w.encode() for w in words], typ)
        for arr in (ls, ls.slice(11, len(ls) // 2)):
            darr = to_device(ls) if arr.offset == 0 else to_device(ls).slice(arr.offset, len(arr))
            for o, p in (combos[:1] + combos[3:] if light else combos):
                got = pc.array_sort_indices(darr, order=o, null_placement=p)
                assert not got.is_cpu and to_host(got).equals(pc.array_sort_indices(arr, order=o, null_placement=p)), (str(typ), o, p)
    tb_dev = pa.table({"b": to_device(bools), "a": dev["i64"]})
    tb_host = pa.table({"b": bools, "a": cols["i64"]})
    for keys in ([("b", "descending"), ("a", "ascending")], [("a", "descending", "at_start"), ("b", "ascending", "at_start")]):
        got = pc.call_function("sort_indices", [tb_dev], pc.SortOptions(sort_keys=keys))        # (by name: the replaced MetaFunction)
        assert to_host(got).equals(pc.sort_indices(tb_host, sort_keys=keys)), keys
    try:
        partition_nth(dev["i64"], n + 1)
        raise SystemExit("a pivot past the end should be refused")
    except pa.ArrowIndexError as e:
        assert "NthToIndices index out of bound" in str(e), e
    assert partition_nth(cols["i64"], 3).type == pa.uint64() and pc.partition_nth_indices(cols["i64"], pivot=3).type == pa.uint64()        # host: the stock kernel
    print("RANK_SELECT_OK")
""")

IMPORT_ORDER_SCRIPT = textwrap.dedent(r"""
    import ctypes, os, sys, faulthandler
    faulthandler.enable()
    import pyarrow as pa          # libarrow and the function registry; pyarrow.compute is NOT imported yet
    assert "pyarrow.compute" not in sys.modules
    sys.path.insert(0, ROOT)
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
    import pyarrow.compute as pc  # its generated wrappers bind the Function objects of the registry as it is NOW
    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(m) for m in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))
    a = pa.array([3, 1, 2, None, 1, 7, None, 2])
    d = to_device(a)
    # the replaced MetaFunctions are what pc.<name> calls: no pc.call_function needed, results in HBM
    r = pc.rank(d, sort_keys="descending", tiebreaker="min")
    assert not r.is_cpu and to_host(r).to_pylist() == [2, 5, 3, 7, 5, 1, 7, 3], to_host(r)
    q = pc.rank_quantile(d)
    assert not q.is_cpu and to_host(q).to_pylist() == [0.5625, 0.125, 0.375, 0.875, 0.125, 0.6875, 0.875, 0.375]
    k = pc.select_k_unstable(d, 3, sort_keys=[("x", "descending")])
    assert not k.is_cpu and to_host(k).to_pylist() == [5, 0, 2]
    p = pc.partition_nth_indices(d, pivot=2)
    assert not p.is_cpu and sorted(to_host(p).to_pylist()) == list(range(8)) and set(to_host(p).to_pylist()[:2]) == {1, 4}
    b = pa.array([True, False, None, True, False, True, False, None])
    keys = [("b", "ascending"), ("a", "descending")]
    s = pc.sort_indices(pa.table({"a": d, "b": to_device(b)}), sort_keys=keys)
    want = pc.sort_indices(pa.table({"a": a, "b": b}), sort_keys=keys)        # (host table: the wrapper hands it to the stock function)
    assert not s.is_cpu and to_host(s).equals(want) and want.to_pylist() == [1, 4, 6, 5, 0, 3, 2, 7], (to_host(s), want)
    # host data through the same wrappers: the stock functions
    assert pc.rank(a, tiebreaker="dense").to_pylist() == [3, 1, 2, 5, 1, 4, 5, 2]
    print("IMPORT_ORDER_OK")
""")

PARQUET_LISTS_SCRIPT = textwrap.dedent(r"""
    import ctypes, faulthandler, os, sys, tempfile
    import numpy as np
    import pyarrow as pa, pyarrow.parquet as pq
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    SC = lambda x: max(64, int(x * float(os.environ.get("ARROW_AMD_TEST_SCALE", "1"))))
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":      # CPU tier: the shim on the emulated kernels (tests/emu)
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin())
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    lib.arrow_amd_parquet_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(n) for n in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    def read_column(path, rg, col):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        rc = lib.arrow_amd_parquet_read_column(path.encode(), rg, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))

    from tests.test_parquet import LIST_VARIANTS, _list_table
    n = SC(300_000)
    for vi, variant in enumerate(LIST_VARIANTS):
        t = _list_table(np.random.default_rng(50 + vi), n, 0.15 if vi % 2 == 0 else 0.0)
        path = os.path.join(tempfile.mkdtemp(), "lists.parquet")
        pq.write_table(t, path, row_group_size=n // 2 + 5, **variant)
        pf = pq.ParquetFile(path)
        tops = [pf.metadata.schema.column(i).path.split(".")[0] for i in range(pf.metadata.num_columns)]
        for rg in range(pf.metadata.num_row_groups):
            ref = pf.read_row_group(rg)
            for ci, top in enumerate(tops):
                d = read_column(path, rg, ci)
                assert not d.is_cpu or len(d) == 0, top
                h = to_host(d)
                h.validate(full=True)
                w = ref.column(top).combine_chunks()
                assert h.type == w.type and h.null_count == w.null_count and h.equals(w), (variant, rg, top, h.type, w.type, h.slice(0, 5), w.slice(0, 5))
    # struct columns of primitives by their top-level FIELD (arrow_amd_parquet_read_field): members as flat leaves under the
    # struct's definition level, the struct's validity from a member's levels; flat and list fields through the same entry
    from tests.test_parquet import _struct_table
    lib.arrow_amd_parquet_read_field.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    def read_field(path, rg, field):
        c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
        rc = lib.arrow_amd_parquet_read_field(path.encode(), rg, field, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    for vi, (sp, mp) in enumerate(((0.15, 0.2), (0.0, 0.2), (0.3, 0.0))):
        t = _struct_table(np.random.default_rng(70 + vi), n, sp, mp)
        path = os.path.join(tempfile.mkdtemp(), "structs.parquet")
        variant = dict(LIST_VARIANTS[vi])
        if isinstance(variant.get("use_dictionary"), list):
            variant["use_dictionary"] = True
        pq.write_table(t, path, row_group_size=n // 2 + 5, **variant)
        pf = pq.ParquetFile(path)
        for rg in range(pf.metadata.num_row_groups):
            ref = pf.read_row_group(rg)
            for fi, name in enumerate(t.schema.names):
                h = to_host(read_field(path, rg, fi))
                h.validate(full=True)
                w = ref.column(name).combine_chunks()
                assert h.type == w.type and h.null_count == w.null_count and h.equals(w), ("struct", vi, rg, name, h.type, w.type)
    lists_path = os.path.join(tempfile.mkdtemp(), "lists2.parquet")
    lt = _list_table(np.random.default_rng(9), SC(20_000), 0.1)
    pq.write_table(lt, lists_path)
    for fi, name in enumerate(lt.schema.names):
        assert to_host(read_field(lists_path, 0, fi)).equals(pq.read_table(lists_path).column(name).combine_chunks()), name
    # what is not a chain of lists over one primitive is refused by name, before any device work
    path = os.path.join(tempfile.mkdtemp(), "s.parquet")
    pq.write_table(pa.table({"s": pa.array([{"a": 1, "b": [1, 2]}, None, {"a": None, "b": []}])}), path)
    c_dev, c_schema = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72)
    for col in (0, 1):
        rc = lib.arrow_amd_parquet_read_column(path.encode(), 0, col, ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert rc != 0 and b"NotImplemented" in lib.arrow_amd_plugin_last_error(), lib.arrow_amd_plugin_last_error()
    print("PARQUET_LISTS_OK")
""")

CASES = [
    ('pyarrow_compute_dispatches_to_the_hip_kernels', SCRIPT, 'PLUGIN_OK', 0.04,
     ''),
    ('device_resident_arrays_through_callfunction', DEVICE_SCRIPT, 'DEVICE_OK', 0.025,
     "SURVEY.md 8 (f1): pyarrow arrays whose buffers live in HBM (kROCM MemoryManager of the plugin, imported through the C Device Data interface) go through Arrow's own CallFunction to the HIP kernels with no staging; results stay on the device and equal the stock CPU results."),
    ('reference_golden_vectors_through_callfunction', GOLDEN_SCRIPT, 'GOLDEN_OK', 1,
     "SURVEY.md 8(c): the reference's own known-answer tests for sort_indices (vector_sort_test.cc:640-724) and the SumOnly group-by (acero/hash_aggregate_test.cc:839-883), replayed through unmodified pyarrow.compute / Acero with the plugin registered — host arrays and device-resident arrays, every key type the path registers."),
    ('acero_fused_group_by_node', ACERO_SCRIPT, 'ACERO_OK', 0.03,
     'SURVEY.md 8(b) "whole-operator replacement": the exec-node factory `aggregate_rocm` drives the fused device group-by from an ordinary Acero plan; results equal Table.group_by\'s.'),
    ('acero_plan_over_a_device_resident_table', ACERO_DEVICE_SCRIPT, 'ACERO_DEVICE_OK', 0.1,
     'SURVEY.md 8 (f2): an ordinary Acero plan table_source -> filter(w > 10) -> project(k, v + w) -> aggregate_rocm over a table whose columns live in HBM. FilterNode (acero/filter_node.cc:73-108) evaluates the expression through ExecuteScalarExpression (compute/expression.cc:722-798) and calls Filter per column, ProjectNode evaluates `add`; every kernel they reach is one of ours and no batch leaves the device until the (small) group-by result.'),
    ('boolean_values_filter_and_take_on_device_resident_arrays', BOOLEAN_VALUES_SCRIPT, 'BOOLEAN_VALUES_OK', 0.025,
     "Filter / take of BOOLEAN (bit-packed) device values through Arrow's CallFunction (arx_take_bits behind the array_filter / array_take shims), incl. sliced operands and EMIT_NULL."),
    ('parquet_column_chunks_through_the_plugin', PARQUET_SCRIPT, 'PARQUET_OK', 0.03,
     "SURVEY.md 8 (f4): parquet::PageReader (headers, decompression) + the C-ABI kernels (levels, indices, dictionary gather, null expansion) -> device-resident arrays equal to the reference's reader."),
    ('parquet_list_columns_through_the_plugin', PARQUET_LISTS_SCRIPT, 'PARQUET_LISTS_OK', 0.01,
     "SURVEY.md 8 (f4), VERDICT r5 missing 5: repeated Parquet columns — list<T> and list<list<T>> of int64 / utf8 / float64 / bool / required int32, null lists, empty lists, null elements, data pages V1 / V2 — through arrow_amd_parquet_read_column: the reference's SchemaManifest for the LevelInfo of every level, DefRepLevelsToList as a kernel over the decoded levels (arx_def_rep_levels_to_list), device-resident ListArrays equal to the reference reader's; structs refused by name."),
    ('single_sync_filter_path_on_device_resident_arrays', MORSEL_FILTER_SCRIPT, 'MORSEL_FILTER_OK', 0.05,
     'The opt-in single-synchronisation device filter (arrow_amd_plugin_set_filter_morsel_rows): worst-case allocation, count -> compact back to back, one read-back — identical output to the default path and to the reference.'),
    ('filter_and_take_of_device_resident_batches_and_tables', SELECTION_META_SCRIPT, 'SELECTION_META_OK', 0.02,
     'FilterMetaFunction / TakeMetaFunction shapes (F4 / T4 of SURVEY.md 8a: record batch, table, chunked array) over device-resident data: per-column array_filter / array_take in HBM, equal to the reference on the host copies; multi-chunk device columns and uncovered device casts are refused instead of read from the CPU.'),
    ('divide_on_device_resident_arrays', DIVIDE_SCRIPT, 'DIVIDE_OK', 0.02,
     'divide / divide_checked (int64, double) on device arrays through CallFunction: values, validity, and the error the last failing valid slot names ("divide by zero" / "overflow"), equal to the reference; `/` in an Acero projection.'),
    ('compare_and_arithmetic_on_every_numeric_type_through_callfunction', NUMERIC_OPS_SCRIPT, 'NUMERIC_OPS_OK', 0.01,
     "The comparison family and add / subtract / multiply (+ _checked) for int8 ... uint32, uint64 and float through CallFunction on device-resident arrays (array x array, array x scalar, slices): results stay in HBM and equal the reference's on the host copies — type, values, validity, null count; the type's own overflow wraps / fails with the reference's text; an Acero filter + projection over int32 / float32 device columns."),
    ('scalar_aggregates_on_device_resident_columns', AGGREGATE_SCRIPT, 'AGGREGATE_OK', 0.02,
     "SumImpl / CountImpl / MinMaxImpl (aggregate_basic.inc.cc:49-110,776-860) as ScalarAggregateKernel shims: `sum`, `count`, `min_max`, `min`, `max` of int64 device columns with every option combination, chunked input (state merge), a refused host+device mix, and Acero's key-less `aggregate` node over a filtered device table."),
    ('acero_order_by_over_a_device_resident_table', ORDER_BY_SCRIPT, 'ORDER_BY_OK', 0.004,
     'SURVEY.md 8 (f2): OrderByNode (acero/order_by_node.cc:100-108) as `order_by_rocm`: table_source -> [filter] -> order_by_rocm over device-resident and host tables, one to three sort keys with their own direction and null placement (int32 / int64 / float64 with NaNs / timestamp), payload columns of int64, utf8, boolean; equal to the stock `order_by` over the host table, with and without threads.'),
    ('parquet_delta_and_split_encodings_through_the_plugin', PARQUET_ENCODINGS_SCRIPT, 'PARQUET_ENCODINGS_OK', 0.02,
     "DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY and BYTE_STREAM_SPLIT column chunks through arrow_amd_parquet_read_column (parquet::PageReader for the pages, the C-ABI kernels for the values), equal to the reference's reader."),
    ('hash_count_min_max_mean_vtables_under_the_stock_group_by_node', HASH_KERNELS_SCRIPT, 'HASH_KERNELS_OK', 0.02,
     'hash_count (three CountOptions modes, any value type with a physical bitmap) / hash_min / hash_max / hash_mean as HashAggregateKernel vtables: the STOCK GroupByNode (pyarrow Table.group_by, Acero "aggregate") lands on the HIP kernels — results equal to the reference kernels\' taken before registration, single- and multi-threaded (Merge), host and device-resident value columns; hash_mean keeps the reference kernel for host values and refuses device values whose partial sums pass 2^53.'),
    ('unique_value_counts_dictionary_encode_drop_null_nonzero_and_numeric_casts_on_device_arrays', VECTOR_HASH_SCRIPT, 'VECTOR_HASH_OK', 0.01,
     "pc.unique / value_counts / dictionary_encode (MASK and ENCODE) through the device Grouper, pc.drop_null, pc.indices_nonzero and every numeric cast pair on device-resident arrays — equal to the reference kernels' results on the same values (slices, several chunks, floats compared by bits, all-null / empty / null-first arrays), outputs stay in HBM, host arrays still reach the reference kernels."),
    ('aggregate_rocm_with_int64_and_multi_column_keys_through_the_device_grouper', GENERAL_GROUP_BY_SCRIPT, 'GENERAL_GROUP_BY_OK', 0.01,
     'aggregate_rocm beyond the fused int32 -> int64 operator: int64 keys over their whole range, 2- and 3-column keys (int32 / int16 / uint8 / date32, nulls as key values), several value columns, sum / mean / min / max / count (three modes) / count_all — the device Grouper + dense hash kernels, equal to the reference GroupByNode with the reference kernels (taken before registration); host batches and device-resident sliced batches.'),
    ('device_streams_sync_events_buffer_reader_writer_and_dlpack', DEVICE_INTERFACES_SCRIPT, 'DEVICE_INTERFACES_OK', 1,
     "SURVEY 8 (f1) completeness: Device::MakeStream / SyncEvent over hipStream_t / hipEvent_t, ArrowDeviceArray.sync_event exported by a producer that does not synchronise and waited for (on the stream, not the host) before the shims' kernels read the imported buffers, MemoryManager::GetBufferWriter / GetBufferReader, DLPack export of a device array into PyTorch-ROCm: zero-copy (pointer equality), refused for nulls / non-numeric types like the reference."),
    ('table_source_rocm_delivers_whole_chunks_to_the_stock_filter_and_project_nodes', TABLE_SOURCE_SCRIPT, 'TABLE_SOURCE_OK', 0.1,
     "table_source_rocm: TableSourceNode without SourceNode's 32Ki-row slicing — the stock FilterNode / ProjectNode then run ONCE per chunk of a device table through the registered kernels (counted), rows / values / order equal to the reference plan's taken before registration; aggregate_rocm consumes such batches where they lie (no staging copy), small and large batches mixed; the knob that replaces the options' default batch size; host tables; validation."),
    ('run_end_encoded_filter_masks_on_device_arrays', REE_FILTER_SCRIPT, 'REE_FILTER_OK', 0.02,
     "array_filter with a run_end_encoded<int16/32/64, boolean> filter over device-resident values: the runs are expanded on the device (arx_ree_bool_expand) and the ordinary filter kernels run — DROP and EMIT_NULL, null run values, logical slices of the REE array, three value widths; equal to the reference's REE filter taken before registration; host arrays keep the reference kernel; mixed residency refused by name."),
    ('reference_golden_grouped_aggregates_through_acero', GOLDEN_HASH_AGGREGATE_SCRIPT, 'GOLDEN_HASH_AGGREGATE_OK', 1,
     'SURVEY.md 8(c): the reference\'s own known-answer tests for the grouped aggregates — acero/hash_aggregate_test.cc CountOnly :714, MeanOnly :959, MeanOverflow :1048, MinMaxOnly :1591, MinMaxTypes :1661, AnyAndAll :2071, AnyAllSlicedNullableBoolean :2160, CountAndSum :3293, SumMeanProductKeepNulls :3481 (tests/golden/reference_vectors.json) — replayed with the plugin registered: Table.group_by and the stock "aggregate" node over host tables (serial and threaded) and over device-resident value columns, and aggregate_rocm over host and device-resident batches.'),
    ('reference_golden_compare_and_arithmetic_through_callfunction', GOLDEN_SCALAR_OPS_SCRIPT, 'GOLDEN_SCALAR_OPS_OK', 1,
     'SURVEY.md 8(c): the reference\'s own known-answer tests for the comparison family (kernels/scalar_compare_test.cc :251-456, every numeric type + timestamps) and add / subtract / multiply / divide with their checked forms (kernels/scalar_arithmetic_test.cc:562-939: wrap-around, "overflow", overflow hidden under a null, "divide by zero", min / -1, signed zeros, null scalars) — tests/golden/reference_vectors_scalar.json — replayed through unmodified pyarrow.compute on device-resident arrays with the plugin registered; bit for bit the stock build\'s answers.'),
    ('reference_kernels_of_the_extended_functions_refuse_device_arrays', DEVICE_GUARD_SCRIPT, 'DEVICE_GUARD_OK', 1,
     "plugin/device_guard.inc: every reference kernel of a function the shim appends kernels to is re-registered behind a check of its operands — a device-resident array of a type without a device kernel (strings under `equal`, booleans under `unique`, decimals under `add`, large_utf8 under `filter`, ...) is refused by name instead of being read by a CPU kernel; host arrays of every type keep the reference's results and errors."),
    ('hash_min_max_of_floats_and_temporal_types', FLOAT_EXTREMA_SCRIPT, 'FLOAT_EXTREMA_OK', 0.02,
     "VERDICT r3 missing 4: hash_min / hash_max / hash_min_max of float32 / float64 / temporal values — the vtables under the stock GroupByNode (host and device-resident values) and aggregate_rocm, equal to the reference's."),
    ('scalar_aggregates_of_float_boolean_and_temporal_device_columns', FLOAT_AGGREGATE_SCRIPT, 'FLOAT_AGGREGATE_OK', 0.02,
     "VERDICT r3 missing 5: sum / mean / min_max / min / max of float / boolean / temporal device-resident columns equal the reference's results bit for bit (the float sum included: the same pairwise summation tree)."),
    ('fill_null_on_device_resident_arrays', FILL_NULL_SCRIPT, 'FILL_NULL_OK', 0.02,
     "VERDICT r3 missing 5: fill_null (= coalesce of two operands) of every fixed-width type on device-resident arrays equals the reference's result; other types are refused by name."),
    ('wrap_device_memory_zero_copy_and_uint64_row_numbers', WRAP_SCRIPT, 'WRAP_OK', 0.02,
     'arrow_amd_wrap_device_memory (caller-owned HBM as a device-resident pyarrow array, no copy) and indices_nonzero writing its uint64 row numbers in one pass.'),
    ('stock_acero_plans_land_on_the_plugin_nodes_after_the_factory_override', ACERO_OVERRIDE_SCRIPT, 'ACERO_OVERRIDE_OK', 0.05,
     'VERDICT r3 weak 9: table_source -> filter -> project -> aggregate (stock names) over a device-resident table.'),
    ('filter_and_take_of_large_utf8_and_large_binary_on_device_arrays', LARGE_BINARY_SCRIPT, 'LARGE_BINARY_OK', 0.02,
     'VERDICT r3 missing 6: large_utf8 / large_binary (int64 offsets) filter, take and drop_null on device-resident arrays.'),
    ('filter_and_take_of_fixed_size_list_and_list_on_device_arrays', NESTED_SELECTION_SCRIPT, 'NESTED_SELECTION_OK', 0.02,
     'VERDICT r3 missing 6: fixed_size_list / list / large_list (fixed-width nested values without nulls) filter, take and drop_null on device-resident arrays; child nulls and other children are refused by name.'),
    ('hash_sum_and_mean_of_floats_are_the_references_row_order_sums', FLOAT_GROUPED_SUM_SCRIPT, 'FLOAT_GROUPED_SUM_OK', 0.01,
     "hash_sum / hash_mean of float32 / float64 — the reference's row-order double accumulation per group, bit for bit, under the stock GroupByNode (host and device-resident values, several batches) and in aggregate_rocm."),
    ('hash_count_distinct_in_aggregate_rocm', COUNT_DISTINCT_SCRIPT, 'COUNT_DISTINCT_OK', 0.02,
     "hash_count_distinct through aggregate_rocm (a second device Grouper over (value, group id) pairs), host and device-resident tables, the three CountOptions modes, fixed-width value types; equal to the reference's GroupByNode."),
    ('hash_sum_of_decimal128_and_decimal_sort_keys', DECIMAL_SUM_SCRIPT, 'DECIMAL_SUM_OK', 0.008,
     'hash_sum of decimal128 columns — 128-bit sums modulo 2^128 on the device, the output widened to precision 38 — under the stock GroupByNode (host and device-resident values, batches, threads) and in aggregate_rocm; hash_mean; hash_min / hash_max of decimal128 in aggregate_rocm; the scalar sum / mean / min_max / min / max of decimal128 device columns.'),
    ('aggregate_rocm_with_key_rows_wider_than_16_bytes', WIDE_KEYS_SCRIPT, 'WIDE_KEYS_OK', 0.004,
     'aggregate_rocm over 18- to 37-byte key rows and a 10-column key: the chain of Grouper tables behind the same node, host and device-resident batches, equal to the reference GroupByNode with the reference kernels.'),
    ('aggregate_rocm_with_utf8_and_binary_keys', STRING_KEYS_SCRIPT, 'STRING_KEYS_OK', 0.005,
     "aggregate_rocm over utf8 / binary key columns (alone, beside fixed-width keys, several of them): the string enters the chain of Grouper tables as its length and 12-byte chunks (arx_binary_key_lengths / _chunk), the unique strings are the strings of the groups' first rows (arx_group_first_rows + the binary take) — equal to the reference GroupByNode, strings that differ only in their last byte, only in length, in trailing NUL bytes, empty vs null."),
    ('stock_group_by_over_device_resident_key_columns_is_served_or_refused', ACERO_GUARD_SCRIPT, 'ACERO_GUARD_OK', 0.05,
     'VERDICT r4 item 8: table_source -> aggregate plans by their STOCK names (what Table.group_by builds) over a table whose KEY columns live in HBM return the reference\'s result (built as aggregate_rocm by the guard arrow_amd_register() installs in front of the CPU Grouper) or a NotImplemented Status; host keys over device values keep the stock GroupByNode; host tables untouched.'),
    ('hash_first_last_one_product_list_distinct_min_max_in_aggregate_rocm', FIRST_LAST_SCRIPT, 'FIRST_LAST_OK', 0.01,
     'VERDICT r4 missing 1: hash_first / hash_last (skip_nulls on and off) / hash_one in aggregate_rocm — the row of every group\'s first / last non-null value (arx_group_edge_rows) + one take — for value types of 1 to 16 bytes, and hash_product (wrapping integer products, double products in row order through the float sums\' walkers) hash_first_last (struct), hash_list (values in row order) and hash_distinct (first-appearance order, three CountOptions modes) — equal to the reference\'s GroupByNode; batches in batch.index order whatever the thread count.'),
    ('rank_select_k_and_partition_nth_on_device_resident_arrays', RANK_SELECT_SCRIPT, 'RANK_SELECT_OK', 0.003,
     "SURVEY.md 8 (f3), VERDICT r5 missing 4: rank (min / max / first / dense), rank_quantile, select_k_unstable and partition_nth_indices by their stock names on device-resident arrays, chunked arrays and tables — the registered HIP sort plus arx_rank's walk of the sorted order; ranks bit for bit the reference's (NaNs, nulls, signed zeros, temporal types), select_k / partition_nth by the property they promise; host data untouched; unsupported device types refused with a Status."),
    ('registered_before_pyarrow_compute_is_imported_the_generated_wrappers_bind_the_replaced_functions', IMPORT_ORDER_SCRIPT, 'IMPORT_ORDER_OK', 1,
     "pyarrow.compute's generated wrappers (pc.rank, pc.select_k_unstable, pc.sort_indices, ...) keep the Function objects they find when the module is imported: with arrow_amd_register() called BEFORE `import pyarrow.compute` they bind the replaced MetaFunctions and device-resident arrays go through them by their ordinary spelling; registered later, the replaced functions are reached by name (CallFunction / pc.call_function) — INTEGRATION.md 'Load order'."),
    ('hash_variance_stddev_skew_kurtosis_in_aggregate_rocm', MOMENTS_SCRIPT, 'MOMENTS_OK', 0.004,
     "SURVEY.md 8 (f3): the grouped moments (GroupedStatisticImpl) as two passes over all rows of the node — null exactly where the reference's Finalize leaves a group null (ddof, unbiased skew / kurtosis of too few values, min_count, skip_nulls), values within 1e-11 relative of the reference's per-batch moments merged batch by batch (its own tests compare approximately)."),
]
