"""The reference's own known-answer tests for the comparison and arithmetic kernels
(tests/golden/reference_vectors_scalar.json; every entry cites kernels/scalar_compare_test.cc or
kernels/scalar_arithmetic_test.cc) as a stream of calls: `cases(gold, ...)` yields one dict per (case, type) with the
function name, the two operands as pyarrow arrays / scalars and the expectation; `check(case, run)` runs it through a
caller-supplied `run(function_name, left, right)` (stock pyarrow.compute, the registered plugin over device-resident
arrays, the oracle) and compares like the reference's AssertBinop / ValidateCompare do — approximately for floating
results (the reference writes 0.32 for 0.64f / 2), exactly for everything else, the sign of a zero included.

Test infrastructure only (needs pyarrow and numpy, nothing else)."""
import datetime
import math

import numpy as np
import pyarrow as pa

SIGNED = ("int8", "int16", "int32", "int64")
UNSIGNED = ("uint8", "uint16", "uint32", "uint64")
FLOATING = ("float32", "float64")
TYPE_SETS = {"integral": SIGNED + UNSIGNED, "signed": SIGNED, "unsigned": UNSIGNED, "floating": FLOATING}
NULL_SCALAR = "null_scalar"


def _resolve(x, name):
    """A JSON value -> a python number of the type under test (None stays None)."""
    if x is None or not isinstance(x, str):
        return x
    if x in ("Inf", "-Inf", "NaN", "-0.0"):
        return float(x.lower().replace("inf", "inf"))
    info = np.iinfo(name)
    lo, hi = int(info.min), int(info.max)
    table = {"min": lo, "max": hi, "max-1": hi - 1, "min+1": lo + 1, "min+2": lo + 2,
             "max/2": hi // 2, "min/2": -((-lo) // 2)}         # (C++ integer division truncates towards zero)
    return table[x]


def _operand(x, name, valid=None):
    typ = getattr(pa, name)()
    if isinstance(x, str) and x == NULL_SCALAR:
        return pa.scalar(None, typ)
    if not isinstance(x, list):
        return pa.scalar(_resolve(x, name), typ)
    vals = [_resolve(v, name) for v in x]
    if valid is not None:      # TweakValidityBit: the values stay in the buffer under the cleared bits
        data = np.array(vals, dtype=name)
        return pa.array(data, type=typ, mask=~np.array(valid, dtype=bool))
    return pa.array(vals, typ)


def _want(x, name):
    typ = getattr(pa, name)()
    if not isinstance(x, list):
        return pa.scalar(_resolve(x, name), typ)
    return pa.array([_resolve(v, name) for v in x], typ)


def cases(gold, types=None, ops=None, scalar_scalar=True):
    """types: the numeric type names to run (default: all ten); ops: function names without "_checked" (default: all).
    scalar_scalar=False drops the cases whose operands are both scalars (they never reach a device kernel)."""
    def type_ok(name):
        return types is None or name in types

    def op_ok(op):
        return ops is None or op in ops

    cmp_ = gold["compare_numeric"]
    for name in SIGNED + UNSIGNED + FLOATING:
        if not type_ok(name):
            continue
        one = cmp_["scalar"]
        for form in ("array_scalar", "scalar_array"):
            for op, rows in cmp_[form].items():
                if not op_ok(op):
                    continue
                for arr, want in rows:
                    l, r = (arr, one) if form == "array_scalar" else (one, arr)
                    yield dict(id=f"compare {form} {op} {name} {arr}", fn=op, left=_operand(l, name), right=_operand(r, name),
                               want=pa.array([None if w is None else bool(w) for w in want], pa.bool_()), raises=None, cite=cmp_["cite"])
        if op_ok("equal"):
            for row in cmp_["null_scalar"]:
                yield dict(id=f"compare null_scalar {name} {row}", fn="equal", left=_operand(row["left"], name), right=_operand(row["right"], name),
                           want=pa.array([None if w is None else bool(w) for w in row["want"]], pa.bool_()), raises=None, cite=cmp_["cite"])
        for row in cmp_["array_array"]:
            if op_ok(row["op"]):
                yield dict(id=f"compare array_array {name} {row}", fn=row["op"], left=_operand(row["left"], name), right=_operand(row["right"], name),
                           want=pa.array([None if w is None else bool(w) for w in row["want"]], pa.bool_()), raises=None, cite=cmp_["cite"])
    if types is None or "timestamp" in types:
        ts = gold["compare_timestamps"]
        epoch = datetime.date(1970, 1, 1)
        secs = [[(datetime.date.fromisoformat(d) - epoch).days * 86400 for d in ts[side]] for side in ("left", "right")]
        for typ in (pa.timestamp("s"), pa.timestamp("s", tz="utc")):
            for op, want in ts["want"].items():
                if op_ok(op):
                    yield dict(id=f"compare timestamps {op} {typ}", fn=op, left=pa.array(secs[0], typ), right=pa.array(secs[1], typ),
                               want=pa.array([bool(w) for w in want], pa.bool_()), raises=None, cite=ts["cite"])
    for c in gold["arithmetic"]:
        if not op_ok(c["op"]):
            continue
        both_scalars = not isinstance(c["left"], list) and not isinstance(c["right"], list)
        if both_scalars and not scalar_scalar:
            continue
        for name in TYPE_SETS[c["types"]]:
            if not type_ok(name):
                continue
            for checked in c["checked"]:
                yield dict(id=f"arith {c['op']}{'_checked' if checked else ''} {name} {c['left']} {c['right']}",
                           fn=c["op"] + ("_checked" if checked else ""),
                           left=_operand(c["left"], name, c.get("left_valid")), right=_operand(c["right"], name, c.get("right_valid")),
                           want=None if "raises" in c else _want(c["want"], name), raises=c.get("raises"), cite=c["cite"])


def as_list(x):
    return x.to_pylist() if isinstance(x, (pa.Array, pa.ChunkedArray)) else [x.as_py()]


def _same_value(a, b, approx, rel):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, float) or isinstance(b, float):
        a, b = float(a), float(b)
        if math.isnan(a) or math.isnan(b):
            return math.isnan(a) and math.isnan(b)
        if a == 0.0 and b == 0.0:
            return math.copysign(1.0, a) == math.copysign(1.0, b)
        if approx:
            return math.isclose(a, b, rel_tol=rel, abs_tol=0.0)
    return a == b


def matches(got, want, approx=True):
    """got / want: pyarrow arrays or scalars of the same type."""
    rel = 1e-6 if want.type == pa.float32() else 1e-12
    g, w = as_list(got), as_list(want)
    return got.type == want.type and len(g) == len(w) and all(_same_value(a, b, approx, rel) for a, b in zip(g, w))


def check(case, run):
    """Runs one case; returns the result (None when the case expects an error) for bit-exact comparisons between runners."""
    if case["raises"] is not None:
        try:
            got = run(case["fn"], case["left"], case["right"])
        except (pa.ArrowInvalid, pa.ArrowNotImplementedError) as e:
            assert isinstance(e, pa.ArrowInvalid) and case["raises"] in str(e), (case["id"], case["cite"], str(e))
            return None
        raise AssertionError((case["id"], case["cite"], "did not fail", as_list(got)))
    got = run(case["fn"], case["left"], case["right"])
    assert matches(got, case["want"]), (case["id"], case["cite"], as_list(got), as_list(case["want"]))
    return got


def cast_cases(gold):
    """The cast_numeric section (kernels/scalar_cast_test.cc:269-431): one dict per case with the input array (nulls
    masked over live values, sliced as the test does), the target type, the CastOptions fields, and `want` (an array)
    or `fails`."""
    for c in gold["cast_numeric"]["cases"]:
        frm, to = getattr(pa, c["from"])(), getattr(pa, c["to"])()
        vals = c["values"]
        if "mask_nulls_at" in c:
            mask = np.zeros(len(vals), dtype=bool)
            mask[c["mask_nulls_at"]] = True
            arr = pa.array(np.array(vals, dtype=c["from"]), type=frm, mask=mask)
        else:
            arr = pa.array(vals, frm)
        if "slice" in c:
            arr = arr.slice(*c["slice"])
        yield dict(id=f"cast {c['from']} -> {c['to']} {vals} {c.get('options', {})}", cite=c["cite"], input=arr, to=to,
                   options=c.get("options", {}), fails=c.get("fails", False),
                   want=None if c.get("fails") else pa.array(c["want"], to))


def check_cast(case, run):
    """run(array, target type, **CastOptions fields) -> array, or raises pa.ArrowInvalid.  Returns the result, or the
    error text for the failing cases (so that two runners can be held to the same message)."""
    if case["fails"]:
        try:
            got = run(case["input"], case["to"], **case["options"])
        except pa.ArrowInvalid as e:
            return str(e)
        raise AssertionError((case["id"], case["cite"], "did not fail", as_list(got)))
    got = run(case["input"], case["to"], **case["options"])
    assert matches(got, case["want"], approx=False), (case["id"], case["cite"], as_list(got), as_list(case["want"]))
    return got


def aggregate_cases(gold, types=None):
    """The scalar_aggregates section (kernels/aggregate_test.cc): one dict per (case, type) with fn, the input as a list
    of chunks (pyarrow arrays of the type under test), the options (a pyarrow options object or None) and `want` (a
    pyarrow scalar: int64 / uint64 / double sums, int64 counts, double means, {min, max} structs)."""
    import pyarrow.compute as pc

    sec = gold["scalar_aggregates"]
    names = [n for n in SIGNED + UNSIGNED + FLOATING if types is None or n in types]

    def opts(o):
        return None if o is None else pc.ScalarAggregateOptions(**o)

    for name in names:
        typ = getattr(pa, name)()
        is_f, is_u = name in FLOATING, name in UNSIGNED
        sum_type = pa.float64() if is_f else pa.uint64() if is_u else pa.int64()
        for c in sec["sum"]:
            yield dict(id=f"sum {name} {c['chunks']} {c['options']}", cite=c["cite"], fn="sum", chunks=[pa.array(x, typ) for x in c["chunks"]],
                       type=typ, options=opts(c["options"]), want=pa.scalar(c["want"], sum_type))
        for c in sec["count"]:
            for mode, want in (("only_valid", c["want"][0]), ("only_null", c["want"][1]), ("all", sum(c["want"]))):
                yield dict(id=f"count {mode} {name} {c['values']}", cite=c["cite"], fn="count", chunks=[pa.array(c["values"], typ)], type=typ,
                           options=pc.CountOptions(mode=mode), want=pa.scalar(want, pa.int64()))
        for c in sec["mean"]:
            want = float("nan") if c["want"] == "NaN" else c["want"]
            yield dict(id=f"mean {name} {c['chunks']} {c['options']}", cite=c["cite"], fn="mean", chunks=[pa.array(x, typ) for x in c["chunks"]],
                       type=typ, options=opts(c["options"]), want=pa.scalar(want, pa.float64()))
        if not is_f:
            st = pa.struct([("min", typ), ("max", typ)])
            for c in sec["min_max"]:
                w = {"min": None, "max": None} if c["want"] is None else {"min": c["want"][0], "max": c["want"][1]}
                yield dict(id=f"min_max {name} {c['chunks']} {c['options']}", cite=c["cite"], fn="min_max",
                           chunks=[pa.array(x, typ) for x in c["chunks"]], type=typ, options=opts(c["options"]), want=pa.scalar(w, st))


def check_aggregate(case, run):
    """run(fn, chunked_array, options) -> pyarrow scalar."""
    got = run(case["fn"], pa.chunked_array(case["chunks"], case["type"]), case["options"])
    want = case["want"]
    if case["fn"] == "min_max":
        ok = got.type == want.type and got.as_py() == want.as_py()
    else:
        ok = got.type == want.type and _same_value(got.as_py(), want.as_py(), False, 0)
    assert ok, (case["id"], case["cite"], got, want)
    return got


def boolean_cases(gold):
    """The boolean_kleene section (kernels/scalar_boolean_test.cc): (function, [operands], want) — the array x array
    cases as written, and every left array against each boolean scalar on either side, where the expectation is the
    function applied to the scalar broadcast to an array (CheckBooleanScalarArrayBinary) — computed here from the
    Kleene truth table, so that it does not depend on the build under test."""
    sec = gold["boolean_kleene"]
    for c in sec["invert"]:
        yield "invert", [pa.array(c["values"], pa.bool_())], pa.array(c["want"], pa.bool_())

    def kleene(fn, a, b):
        if fn == "and_kleene":
            return False if (a is False or b is False) else (None if (a is None or b is None) else True)
        return True if (a is True or b is True) else (None if (a is None or b is None) else False)

    for fn in ("and_kleene", "or_kleene"):
        for c in sec[fn]:
            left, right = pa.array(c["left"], pa.bool_()), pa.array(c["right"], pa.bool_())
            assert [kleene(fn, a, b) for a, b in zip(c["left"], c["right"])] == c["want"]      # (the table agrees with the transcription)
            yield fn, [left, right], pa.array(c["want"], pa.bool_())
            for sv in (None, True, False):
                sc = pa.scalar(sv, pa.bool_())
                yield fn, [sc, left], pa.array([kleene(fn, sv, a) for a in c["left"]], pa.bool_())
                yield fn, [left, sc], pa.array([kleene(fn, a, sv) for a in c["left"]], pa.bool_())
