"""Replays the reference's own known-answer tests for the grouped aggregates (tests/golden/reference_vectors.json, the
hash_* sections; every entry cites acero/hash_aggregate_test.cc) through a caller-supplied group-by.

`run(table, aggs)` takes a pyarrow Table with a "key" column and a list of (column, function, options) — function
names as Table.group_by(...).aggregate spells them ("sum", "count", "mean", "min", "max", "any", "all", "count_all"),
options None or a plain dict ({"mode"} for count, {"skip_nulls", "min_count"} otherwise; pc_options() converts) — and returns
(key_list, [output_list per aggregate]) in any group order.  The same replay runs against the stock pyarrow build (which
pins the transcription), the registered plugin kernels behind GroupByNode, and the fused aggregate_rocm node.

Test infrastructure only (needs pyarrow, nothing else)."""
import math

import pyarrow as pa
import pyarrow.compute as pc


SECTIONS = ("hash_count_only", "hash_mean_only", "hash_mean_overflow", "hash_min_max_only", "hash_min_max_types", "hash_any_all",
            "hash_any_all_sliced", "hash_count_and_sum", "hash_sum_mean_keep_nulls")


def _sorted_rows(keys, outs):
    rows = [[k] + [o[i] for o in outs] for i, k in enumerate(keys)]
    return sorted(rows, key=lambda r: (r[0] is None, 0 if r[0] is None else r[0]))


def _table(batches, arg_type, key_type, names=("argument",)):
    rbs = [pa.record_batch({**{n: pa.array(b[n], arg_type) for n in names}, "key": pa.array(b["key"], key_type)})
           for b in batches]
    return pa.Table.from_batches(rbs)


def _opts(skip_nulls, min_count):
    return {"skip_nulls": skip_nulls, "min_count": min_count}


def pc_options(opts):
    """The plain-dict options of an aggregate (None, {"mode"} for count, {"skip_nulls", "min_count"}) as pyarrow's objects."""
    if opts is None:
        return None
    return pc.CountOptions(mode=opts["mode"]) if "mode" in opts else pc.ScalarAggregateOptions(**opts)


def _close(a, b, rtol):
    if a is None or b is None:
        return a is None and b is None
    return math.isclose(float(a), float(b), rel_tol=rtol, abs_tol=0.0) if rtol else a == b


def replay(gold, run, key_types=(pa.int64(),), int_types=None, sections=None):
    """Returns the number of (case, type) combinations checked; raises AssertionError with the case on a mismatch."""
    ran = 0

    def want(name):
        return sections is None or name in sections

    for key_type in key_types:
        if want("hash_count_only"):
            c = gold["hash_count_only"]
            t = _table(c["batches"], pa.float64(), key_type)
            for mode, rows in c["want_sorted_by_key"].items():
                keys, outs = run(t, [("argument", "count", {"mode": mode})])
                assert _sorted_rows(keys, outs) == rows, ("hash_count_only", mode, str(key_type), _sorted_rows(keys, outs))
                ran += 1
            keys, outs = run(t, [("argument", "count", None)])      # (default options = ONLY_VALID)
            assert _sorted_rows(keys, outs) == c["want_sorted_by_key"]["only_valid"]
            ran += 1
        if want("hash_mean_only"):
            c = gold["hash_mean_only"]
            t = _table(c["batches"], pa.int64(), key_type)
            for case in c["cases"]:
                keys, outs = run(t, [("argument", "mean", _opts(case["skip_nulls"], case["min_count"]))])
                assert _sorted_rows(keys, outs) == case["want_sorted_by_key"], ("hash_mean_only", case, _sorted_rows(keys, outs))
                ran += 1
        if want("hash_mean_overflow"):
            c = gold["hash_mean_overflow"]
            t = _table(c["batches"], pa.int64(), key_type)
            keys, outs = run(t, [("argument", "mean", None)])
            got = _sorted_rows(keys, outs)
            assert len(got) == len(c["want_sorted_by_key"]) and all(
                g[0] == w[0] and _close(g[1], w[1], c["rtol"]) for g, w in zip(got, c["want_sorted_by_key"])), ("hash_mean_overflow", got)
            ran += 1
        if want("hash_min_max_only"):
            c = gold["hash_min_max_only"]
            t = _table(c["batches"], pa.int64(), key_type)
            keys, outs = run(t, [("argument", "min", None), ("argument", "max", None)])
            assert _sorted_rows(keys, outs) == c["want_sorted_by_key"], ("hash_min_max_only", _sorted_rows(keys, outs))
            ran += 1
        if want("hash_min_max_types"):
            c = gold["hash_min_max_types"]
            for name in c["types"]:
                typ = getattr(pa, name)()
                if int_types is not None and typ not in int_types:
                    continue
                t = _table(c["batches"], typ, key_type)
                keys, outs = run(t, [("argument", "min", None), ("argument", "max", None)])
                assert _sorted_rows(keys, outs) == c["want_sorted_by_key"], ("hash_min_max_types", name, _sorted_rows(keys, outs))
                ran += 1
        if want("hash_any_all"):
            c = gold["hash_any_all"]
            t = _table(c["batches"], pa.bool_(), key_type)
            for case in c["cases"]:
                keys, outs = run(t, [("argument", case["function"], _opts(case["skip_nulls"], case["min_count"]))])
                rows = _sorted_rows(keys, outs)
                assert [r[0] for r in rows] == c["keys_sorted"] and [r[1] for r in rows] == case["want"], ("hash_any_all", case, rows)
                ran += 1
            # all eight at once, as the reference test asks for them (one Grouper, eight kernel states)
            keys, outs = run(t, [("argument", k["function"], _opts(k["skip_nulls"], k["min_count"])) for k in c["cases"]])
            rows = _sorted_rows(keys, outs)
            for j, case in enumerate(c["cases"]):
                assert [r[1 + j] for r in rows] == case["want"], ("hash_any_all together", case, rows)
            ran += 1
        if want("hash_any_all_sliced"):
            c = gold["hash_any_all_sliced"]
            full = pa.table({"any_arg": pa.array(c["any_arg"], pa.bool_()), "all_arg": pa.array(c["all_arg"], pa.bool_()),
                             "key": pa.array(c["key"], key_type)})
            keys, outs = run(full.slice(c["slice_offset"]), [("any_arg", "any", None), ("all_arg", "all", None)])
            assert _sorted_rows(keys, outs) == c["want"], ("hash_any_all_sliced", _sorted_rows(keys, outs))
            ran += 1
        if want("hash_count_and_sum"):
            c = gold["hash_count_and_sum"]
            w = c["want"]
            t = pa.table({"argument": pa.array(c["argument"], pa.int64()), "key": pa.array(c["key"], key_type),
                          "key_copy": pa.array(c["key"], pa.int64())})
            keys, outs = run(t, [("argument", "count", {"mode": "only_valid"}),
                                 ("argument", "count", {"mode": "only_null"}),
                                 ("argument", "count", {"mode": "all"}),
                                 ([], "count_all", None),
                                 ("argument", "sum", None),
                                 ("argument", "sum", _opts(True, 3)),
                                 ("key_copy", "sum", None)])
            expect = _sorted_rows(w["key"], [w["count_only_valid"], w["count_only_null"], w["count_all_mode"], w["count_all"],
                                             w["sum"], w["sum_min_count_3"], w["sum_of_key"]])
            assert _sorted_rows(keys, outs) == expect, ("hash_count_and_sum", _sorted_rows(keys, outs))
            ran += 1
        if want("hash_sum_mean_keep_nulls"):
            c = gold["hash_sum_mean_keep_nulls"]
            w = c["want"]
            t = pa.table({"argument": pa.array(c["argument"], pa.int64()), "key": pa.array(c["key"], key_type)})
            keys, outs = run(t, [("argument", "sum", _opts(False, 1)), ("argument", "sum", _opts(False, 3)),
                                 ("argument", "mean", _opts(False, 1)), ("argument", "mean", _opts(False, 3))])
            expect = _sorted_rows(w["key"], [w["sum"], w["sum_min_count_3"], w["mean"], w["mean_min_count_3"]])
            assert _sorted_rows(keys, outs) == expect, ("hash_sum_mean_keep_nulls", _sorted_rows(keys, outs))
            ran += 1
    return ran


def stock_group_by(use_threads=False):
    """Table.group_by(...).aggregate(...): Acero's GroupByNode with whatever hash_* kernels the registry holds."""
    def run(table, aggs):
        r = table.group_by("key", use_threads=use_threads).aggregate([(c, f, pc_options(o)) for c, f, o in aggs])
        names = r.column_names
        ki = len(names) - 1 - names[::-1].index("key")      # ("key_copy_sum" etc. never equal "key"; the key column is named "key")
        return r.column(ki).to_pylist(), [r.column(i).to_pylist() for i in range(len(names)) if i != ki]
    return run


def declaration_group_by(factory, use_threads=False):
    """The same aggregates through an Acero Declaration: table_source -> <factory> (e.g. "aggregate" or "aggregate_rocm")."""
    from pyarrow import acero

    def run(table, aggs):
        specs = [(col, "hash_" + fn, pc_options(opts), f"out{j}") for j, (col, fn, opts) in enumerate(aggs)]
        r = acero.Declaration.from_sequence([
            acero.Declaration("table_source", acero.TableSourceNodeOptions(table)),
            acero.Declaration(factory, acero.AggregateNodeOptions(specs, keys=["key"])),
        ]).to_table(use_threads=use_threads)
        return r.column("key").to_pylist(), [r.column(f"out{j}").to_pylist() for j in range(len(aggs))]
    return run


def replay_scalar_arguments(gold, run):
    """The hash_scalar_arguments section: `run(case, aggs)` gets the case dict (batches with either a "scalar" or an
    "argument" list beside "key") and the aggregates as (function, options) pairs, and returns (key_list, [output lists])."""
    ran = 0
    for case in gold["hash_scalar_arguments"]["cases"]:
        aggs = [(fn, opts) for fn, opts in case["aggregates"]]
        keys, outs = run(case, aggs)
        got = _sorted_rows(keys, outs)
        want = case["want_sorted_by_key"]
        assert len(got) == len(want) and all(len(g) == len(w) and all(_close(a, b, 1e-15 if isinstance(b, float) else 0) for a, b in zip(g, w))
                                             for g, w in zip(got, want)), (case["name"], got)
        ran += 1
    return ran


def union_of_scalar_batches(factory, use_threads=False):
    """Acero: every batch is its own table_source; the scalar ones go through a ProjectNode whose literal expression
    leaves a SCALAR in the ExecBatch (ExecuteScalarExpression), a union joins them, <factory> aggregates."""
    from pyarrow import acero

    def run(case, aggs):
        typ = pa.bool_() if case["argument_type"] == "bool" else getattr(pa, case["argument_type"])()
        inputs = []
        for b in case["batches"]:
            if "scalar" in b:
                src = acero.Declaration("table_source", acero.TableSourceNodeOptions(pa.table({"key": pa.array(b["key"], pa.int64())})))
                proj = acero.Declaration("project", acero.ProjectNodeOptions([pc.scalar(pa.scalar(b["scalar"], typ)), pc.field("key")],
                                                                             ["argument", "key"]), [src])
                inputs.append(proj)
            else:
                t = pa.table({"argument": pa.array(b["argument"], typ), "key": pa.array(b["key"], pa.int64())})
                inputs.append(acero.Declaration("table_source", acero.TableSourceNodeOptions(t)))
        union = acero.Declaration("union", acero.ExecNodeOptions(), inputs)
        specs = [("argument", "hash_" + fn, pc_options(opts), f"out{j}") for j, (fn, opts) in enumerate(aggs)]
        r = acero.Declaration(factory, acero.AggregateNodeOptions(specs, keys=["key"]), [union]).to_table(use_threads=use_threads)
        return r.column("key").to_pylist(), [r.column(f"out{j}").to_pylist() for j in range(len(aggs))]
    return run
