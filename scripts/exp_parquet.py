"""Parquet -> HBM decode of one file vs pyarrow's reader: total time and the device part alone."""
import os, sys, tempfile, time
import numpy as np, torch
import pyarrow as pa, pyarrow.parquet as pq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import parquet as P
n = int(os.environ.get("N", 20_000_000))
rng = np.random.default_rng(1)
t = pa.table({"k": pa.array(rng.integers(0, 5000, n), mask=rng.random(n) < 0.1),
              "v": pa.array(rng.integers(-2**62, 2**62, n)),
              "w": pa.array(np.cumsum(rng.integers(-3, 4, n)))})       # compressible PLAIN int64: the device Snappy route
t = t.cast(pa.schema([pa.field("k", pa.int64()), pa.field("v", pa.int64()), pa.field("w", pa.int64(), nullable=False)]))
path = os.path.join(tempfile.mkdtemp(), "t.parquet")
pq.write_table(t, path, row_group_size=n, compression="snappy", use_dictionary=["k"],
               data_page_version=os.environ.get("PAGE_VERSION", "1.0"))
print("file MB", os.path.getsize(path) / 1e6, flush=True)
t0 = time.perf_counter(); ref = pq.read_table(path, use_threads=False); t_ref = time.perf_counter() - t0
t0 = time.perf_counter(); ref = pq.read_table(path, use_threads=True); t_ref_mt = time.perf_counter() - t0
amd.parquet.read_table(path); torch.cuda.synchronize()
stats = {}
t0 = time.perf_counter(); got = amd.parquet.read_table(path, stats=stats); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("stats", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()})
P.DEVICE_SNAPPY = False
amd.parquet.read_table(path); torch.cuda.synchronize()
t0 = time.perf_counter(); amd.parquet.read_table(path); torch.cuda.synchronize(); t_host_codec = time.perf_counter() - t0
P.DEVICE_SNAPPY = True
print(f"arrow_amd.parquet.read_table with the host Snappy codec for every page: {t_host_codec*1e3:.0f} ms")
for name in ("k", "v", "w"):
    assert got[name][0].to_pyarrow().equals(ref.column(name).combine_chunks()), name
print(f"pyarrow read_table: {t_ref*1e3:.0f} ms (1 thread), {t_ref_mt*1e3:.0f} ms (threads) | arrow_amd.parquet.read_table: {t_all*1e3:.0f} ms "
      f"(host: metadata, page headers, snappy, run-header walk in Python; device: levels, indices, dictionary gather, expand)")
from arrow_amd import tracing
