"""sum / min_max of a 1B-row int64 column (one streaming pass) vs the reference CPU kernels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd, bench
dev = torch.device("cuda", 0)
n = 1_000_000_000
values, validity, mask, _ = bench.gen_filter_inputs(n, dev, 0, 0.10, 0.10)
a = amd.Array(amd.array.int64, n, [validity, values], -1, 0)
agg = amd.compute.Int64Aggregator(dev)
agg.consume(a); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): agg.consume(a)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"reduce (sum+count+min+max) of 1e9 int64 + validity: {ms:.3f} ms = {8.125 * n / ms / 1e6:.0f} GB/s algorithmic")
import pyarrow as pa, pyarrow.compute as pc, numpy as np
m = 1 << 27
host = pa.array(values[: m * 8].view(torch.int64).cpu().numpy(), mask=~np.unpackbits(validity[: m // 8].cpu().numpy(), bitorder="little").astype(bool))
t = time.perf_counter(); r1 = pc.sum(host); r2 = pc.min_max(host); dt = time.perf_counter() - t
print(f"pyarrow sum + min_max on the first 2^27 rows: {dt*1e3:.1f} ms = {m/dt/1e9:.2f} Grows/s (1 thread)")
part = amd.compute.Int64Aggregator(dev); part.consume(a.slice(0, m))
print("parity:", part.sum() == r1.as_py(), part.min_max() == (r2.as_py()["min"], r2.as_py()["max"]))
