"""Key rows wider than one device Grouper table (16 bytes / 8 columns) and utf8 / binary keys: the chain of tables, and
the one-pass (length, hash) form of var-width keys with its verification and forced-collision fallback.  (Written at the
end of round 3 in a last-sorting file because it had not yet run on gfx950; green there since round 4 —
profiles/r04_c_*, r04_e_* — and named like its siblings since.  The emulator tier runs the same checks.)"""
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from . import parity_cases as P
from .test_gpu_arrow_plugin import ROOT

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


def _run(script, marker):
    pytest.importorskip("pyarrow")
    code = f"ROOT = {ROOT!r}\n" + script
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and marker in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _rng(*key):
    return np.random.default_rng([20260924, *[abs(hash(k)) % (1 << 31) for k in key]])


@pytest.mark.gpu
@pytest.mark.parametrize("dtypes,n,card,null_p,batches", [
    ((np.int64, np.int64, np.int64), 200_000, 5000, 0.1, 2), ((np.int64,) * 5, 100_000, 90_000, 0.05, 3),
    ((np.uint8,) * 20, 150_000, 400, 0.2, 2), ((np.int32, np.int64, np.int16, np.float64, np.int8, np.int64), 100_000, 30, 0.3, 1),
    ((np.int64, np.int64, np.int32), 0, 1, 0.0, 1)])
def test_grouper_rows_wider_than_one_table(gpu_ctx, dtypes, n, card, null_p, batches):
    """arrow_amd.compute.Grouper over rows of 20-40 bytes / 20 columns: ids in order of first appearance, uniques and
    Lookup equal to the oracle's GrouperImpl restatement (row/grouper.cc:695-815, :835-940)."""
    P.check_grouper(gpu_ctx, _rng("wide", len(dtypes), n, card), dtypes, n, card, null_p, batches)


@pytest.mark.gpu
def test_grouper_chain_levels_and_partial_lookups(gpu_ctx):
    P.check_grouper_chain(gpu_ctx)


@pytest.mark.gpu
def test_group_by_three_int64_keys(gpu_ctx):
    P.check_group_by_keys(gpu_ctx, _rng("wide-gb"), (np.int64, np.int64, np.int64), 300_000, 12, 0.1)


@pytest.mark.gpu
def test_aggregate_rocm_with_key_rows_wider_than_16_bytes():
    """aggregate_rocm over 18- to 37-byte key rows and a 10-column key: the chain of Grouper tables behind the same node,
    host and device-resident batches, equal to the reference GroupByNode with the reference kernels."""
    _run(WIDE_KEYS_SCRIPT, "WIDE_KEYS_OK")


@pytest.mark.gpu
def test_aggregate_rocm_with_utf8_and_binary_keys():
    """aggregate_rocm over utf8 / binary key columns (alone, beside fixed-width keys, several of them): the string enters
    the chain of Grouper tables as its length and 12-byte chunks (arx_binary_key_lengths / _chunk), the unique strings are
    the strings of the groups' first rows (arx_group_first_rows + the binary take) — equal to the reference GroupByNode,
    strings that differ only in their last byte, only in length, in trailing NUL bytes, empty vs null."""
    _run(STRING_KEYS_SCRIPT, "STRING_KEYS_OK")


@pytest.mark.gpu
@pytest.mark.parametrize("n,null_p,offset,max_len,card", [(0, 0.0, 0, 8, 1), (200_000, 0.1, 0, 30, 400), (100_000, 0.0, 5, 11, 90_000),
                                                         (50_000, 0.3, 3, 50, 7), (1000, 1.0, 0, 5, 5), (60_000, 0.1, 2, 700, 9000)])
def test_binary_key_columns_and_first_rows(gpu_ctx, n, null_p, offset, max_len, card):
    from . import util as U

    P.check_binary_key_columns(gpu_ctx, U.random_binary_pool(_rng("bkey", n, max_len), n, card, null_p, offset, max_len), _rng("bkey2", n))
