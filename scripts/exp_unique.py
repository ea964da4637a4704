"""unique / value_counts on int32 (fused table + sort of the groups) vs the reference CPU kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
dev = torch.device("cuda", 0)
n = 1 << 28
g = torch.Generator(device=dev).manual_seed(2)
for groups in (1000, 1_000_000, 10_000_000):
    keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
    a = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    cap = 2 * groups + 2
    for name, fn in (("unique", lambda: amd.compute.unique(a, capacity=cap)), ("value_counts", lambda: amd.compute.value_counts(a, capacity=cap))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{name:13s} n=2^28 groups={groups:>9}: {dt*1e3:7.2f} ms = {n/dt/1e9:6.1f} Grows/s", flush=True)
    if groups == 1_000_000:
        import pyarrow as pa, pyarrow.compute as pc
        s = 1 << 25
        host = pa.array(keys[:s].cpu().numpy())
        t = time.perf_counter(); ref = pc.unique(host); dt = time.perf_counter() - t
        print(f"pyarrow pc.unique on the first 2^25 rows: {dt*1e3:.1f} ms = {s/dt/1e9:.3f} Grows/s (1 thread)")
        got = amd.compute.unique(amd.Array(amd.array.int32, s, [None, keys[:s].contiguous().view(torch.uint8)], 0, 0), capacity=cap)
        print("parity vs pyarrow:", got.to_pyarrow().equals(ref))
