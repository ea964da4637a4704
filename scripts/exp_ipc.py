"""IPC file with LZ4_FRAME buffer compression -> HBM: buffers decompressed on the device vs pyarrow's reader (host LZ4) + upload."""
import os, sys, tempfile, time
import numpy as np, torch, pyarrow as pa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
n = int(os.environ.get("N", 20_000_000))
rng = np.random.default_rng(1)
t = pa.table({"k": pa.array(rng.integers(0, 5000, n), mask=rng.random(n) < 0.1),
              "v": pa.array(rng.integers(-2**62, 2**62, n)),
              "w": pa.array(np.cumsum(rng.integers(-3, 4, n)))})
path = os.path.join(tempfile.mkdtemp(), "t.arrow")
with pa.ipc.new_file(path, t.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
    for b in t.to_batches(max_chunksize=int(os.environ.get("BATCH", 1 << 20))):
        w.write_batch(b)
print("file MB", round(os.path.getsize(path) / 1e6, 1), "uncompressed MB", round(t.nbytes / 1e6, 1), flush=True)


def best(fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), r


t_ref, ref = best(lambda: pa.ipc.open_file(pa.memory_map(path, "r")).read_all())
stats = {}
t_dev, got = best(lambda: amd.ipc.read_table(path, stats=stats, device_decompress=True), reps=int(os.environ.get('REPS', 3)))
assert stats["device_lz4_batches"] > 0
t_host, got2 = best(lambda: amd.ipc.read_table(path, device_decompress=False))
for name in ("k", "v", "w"):
    for i, arr in enumerate(got[name]):
        assert arr.to_pyarrow().equals(ref.column(name).chunk(i)), (name, i)
print(f"pyarrow read_all (host LZ4, result on the host): {t_ref*1e3:.0f} ms | arrow_amd.ipc.read_table, buffers decompressed on the device: "
      f"{t_dev*1e3:.0f} ms | the same through pyarrow's reader + upload: {t_host*1e3:.0f} ms", flush=True)
