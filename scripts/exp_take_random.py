"""take with random indices: where its time goes.  10^8 random uint32 indices into a 10^9-row int64
column — with the column's validity bitmap (the bench's leg: one more random probe per index, into 125 MB) and without it,
and with the indices SORTED first (every 128-byte line of the column fetched once, adjacent lanes on the same line):
the floor a bucket-by-line stage could reach if bucketing were free."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402

dev = torch.device("cuda", 0)
n, m = 1_000_000_000, 100_000_000
g = torch.Generator(device=dev).manual_seed(5)
vals = torch.empty(n, dtype=torch.int64, device=dev)
for b in range(0, n, 1 << 27):
    e = min(n, b + (1 << 27))
    vals[b:e] = torch.randint(-2**62, 2**62, (e - b,), dtype=torch.int64, device=dev, generator=g)
validity = torch.randint(0, 256, ((n + 7) // 8 + 8,), dtype=torch.uint8, device=dev, generator=g)
idx = torch.randint(0, n, (m,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
idx_sorted = torch.sort(idx.to(torch.int64))[0].to(torch.int32)


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del out
    return min(ts)


for label, v in (("values with a validity bitmap", validity), ("values without nulls", None)):
    dv = amd.Array(amd.array.int64, n, [v, vals.view(torch.uint8)], -1 if v is not None else 0, 0)
    for ilabel, ix in (("random", idx), ("the same indices sorted", idx_sorted)):
        di = amd.Array(amd.array.uint32, m, [None, ix.view(torch.uint8)], 0, 0)
        ms = timed(lambda: amd.compute.take(dv, di, boundscheck=False))
        alg = 20.25 * m if v is not None else 20.0 * m
        print(f"take {m} uint32 indices ({ilabel}) into {n} int64 rows, {label}: {ms:.3f} ms = {alg / ms / 1e6:.0f} GB/s algorithmic = {alg / ms / 1e6 / 8000:.4f} of 8 TB/s", flush=True)
