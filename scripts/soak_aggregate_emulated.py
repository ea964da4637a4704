"""Random-shape soak of aggregate_rocm over device-resident chunked tables on the EMULATED tier (no GPU): 1-5 chunks, each with its own
null rates (none / some / all) and array offset, random flush thresholds; hash_sum / hash_count / hash_min / hash_max against Acero's own
aggregate on the host copy.  Usage: soak_aggregate_emulated.py <first seed> <trials>."""
import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, pyarrow as pa, pyarrow.acero as acero
from tests.emu.build_plugin_emu import build_plugin
lib = ctypes.CDLL(build_plugin(verbose=False))
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
lib.arrow_amd_plugin_set_aggregate_flush_rows.argtypes = [ctypes.c_int64]
lib.arrow_amd_plugin_set_aggregate_direct_rows.argtypes = [ctypes.c_int64]
lib.arrow_amd_plugin_set_table_source_rows.argtypes = [ctypes.c_int64]
lib.arrow_amd_plugin_set_coalesce_rows.argtypes = [ctypes.c_int64]
assert lib.arrow_amd_register() == 0
def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(n) for n in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
seed0, trials = int(sys.argv[1]), int(sys.argv[2])
for trial in range(trials):
    rng = np.random.default_rng(seed0 + trial)
    nchunks = int(rng.integers(1, 6))
    hk, hv, dk, dv = [], [], [], []
    for c in range(nchunks):
        m = int(rng.integers(1, 90_000))
        kp, vp = float(rng.choice([0, 0, 0.05, 0.5])), float(rng.choice([0, 0, 0.2, 1.0]))
        k = pa.array(rng.integers(-300, 300, m).astype(np.int32), mask=(rng.random(m) < kp) if kp else None)
        v = pa.array(rng.integers(-2**40, 2**40, m), mask=(rng.random(m) < vp) if vp else None)
        start = int(rng.integers(0, min(m, 70)))
        k, v = k.slice(start), v.slice(start)
        hk.append(k); hv.append(v); dk.append(to_device(k)); dv.append(to_device(v))
    host = pa.table({"k": pa.chunked_array(hk), "v": pa.chunked_array(hv)})
    dev = pa.table({"k": pa.chunked_array(dk), "v": pa.chunked_array(dv)})
    # round 3: the source (stock / whole-chunk / stock + coalesce_rocm), the rows above which a batch is consumed in place,
    # the batch size of table_source_rocm and the coalescing target are random too
    source = str(rng.choice(["table_source", "table_source_rocm", "coalesce"]))
    lib.arrow_amd_plugin_set_aggregate_direct_rows(int(rng.choice([1 << 22, 1, 5_000, 40_000])))
    lib.arrow_amd_plugin_set_table_source_rows(int(rng.choice([1 << 27, 7_000, 33_000])))
    lib.arrow_amd_plugin_set_coalesce_rows(int(rng.choice([1 << 26, 20_000, 70_000])))
    def plan(t, agg):
        head = [acero.Declaration("table_source" if agg == "aggregate" or source == "coalesce" else source, acero.TableSourceNodeOptions(t))]
        if agg != "aggregate" and source == "coalesce":
            import pyarrow.compute as pc
            head.append(acero.Declaration("coalesce_rocm", acero.FilterNodeOptions(pc.scalar(True))))
        return acero.Declaration.from_sequence(head + [
            acero.Declaration(agg, acero.AggregateNodeOptions([("v", "hash_sum", None, "s"), ("v", "hash_count", None, "c"), ("v", "hash_min", None, "lo"), ("v", "hash_max", None, "hi")], keys=["k"]))])
    want = plan(host, "aggregate").to_table(use_threads=False).select(["k", "s", "c", "lo", "hi"]).sort_by("k")
    lib.arrow_amd_plugin_set_aggregate_flush_rows(int(rng.choice([1 << 21, 10_000, 50_000, 1])))
    got = plan(dev, "aggregate_rocm").to_table(use_threads=False).select(["k", "s", "c", "lo", "hi"]).sort_by("k")
    assert got.equals(want), (trial, got.slice(0, 5), want.slice(0, 5))
print("AGG_SOAK_OK")
