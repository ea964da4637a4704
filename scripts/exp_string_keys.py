"""aggregate_rocm over a binary key column of FIXED length L (1e7 rows, 1e5 distinct keys, device-resident): the one-pass
(length, 64-bit hash) + verification route against the exact 12-byte chunk chain (hash bits 0), L = 8 / 64 / 512."""
import ctypes, os, sys, time
import numpy as np
import pyarrow as pa, pyarrow.compute as pc
from pyarrow import acero
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_amd.plugin_build import build_plugin
lib = ctypes.CDLL(build_plugin(verbose=False))
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
lib.arrow_amd_plugin_string_key_hash_collisions.restype = ctypes.c_int64
assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()
def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
rng = np.random.default_rng(5)
n0, card = int(os.environ.get("ROWS", 10_000_000)), 100_000
for L in (8, 64, 512):
    n = min(n0, (2**31 - 64) // L)        # (int32 offsets: a binary column holds < 2 GB of bytes)
    ids = rng.integers(0, card, n)
    dv = to_device(pa.array(rng.integers(-2**40, 2**40, n)))
    pool = rng.integers(0, 256, (card, L), dtype=np.uint8)
    pool[:, : L - 4] = pool[0, : L - 4]            # every key shares its first L - 4 bytes
    data = pool[ids].reshape(-1)
    offsets = (np.arange(n + 1, dtype=np.int64) * L).astype(np.int32)
    keys = pa.Array.from_buffers(pa.binary(), n, [None, pa.py_buffer(offsets), pa.py_buffer(data)])
    td = pa.Table.from_batches([pa.RecordBatch.from_arrays([to_device(keys), dv], names=["k", "v"])])
    plan = acero.Declaration.from_sequence([
        acero.Declaration("table_source", acero.TableSourceNodeOptions(td)),
        acero.Declaration("aggregate_rocm", acero.AggregateNodeOptions([("v", "hash_sum", None, "s")], keys=["k"]))])
    for bits in (64, 0):
        lib.arrow_amd_plugin_set_string_key_hash_bits(ctypes.c_int64(bits))
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); out = plan.to_table(use_threads=False); best = min(best, time.perf_counter() - t0)
        assert out.num_rows == card
        print(f"L = {L:4d} bytes, {n} rows, hash bits {bits:2d}: {best * 1e3:9.2f} ms  ({n / best / 1e6:8.1f} Mrows/s)  collisions {lib.arrow_amd_plugin_string_key_hash_collisions()}", flush=True)
