"""Binary (utf8) filter/take timing on the GPU box: n strings of 0..24 bytes built on the device,
10 % mask; per-stage HIP-event times and the pyarrow CPU kernels on a sample beside them."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
A = amd.array
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
n = int(os.environ.get("N", 1 << 27))
lens = torch.randint(0, 25, (n,), dtype=torch.int32, device=dev, generator=g)
offs = torch.zeros(n + 1, dtype=torch.int32, device=dev)
offs[1:] = torch.cumsum(lens, 0, dtype=torch.int32)
total = int(offs[-1])
data = torch.randint(32, 127, (total,), dtype=torch.uint8, device=dev, generator=g)
valid = torch.rand(n, device=dev, generator=g) >= 0.1
mask = torch.rand(n, device=dev, generator=g) < 0.1
def bitmap(b):
    w = b.view(-1, 8).to(torch.uint8) * (1 << torch.arange(8, device=dev, dtype=torch.uint8))
    return w.sum(1, dtype=torch.uint8)
vb, mb = bitmap(valid), bitmap(mask)
vals = A.Array(A.utf8, n, [vb, offs.view(torch.uint8), data], -1, 0)
m = A.Array(A.bool_, n, [None, mb], 0, 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): out = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, out
ms, out = timed(lambda: amd.compute.filter(vals, m))
ob = int(out.buffers[1].view(torch.int32)[out.length])
alg = n / 8 * 2 + out.length * (4 + 4 + 4) + 2 * ob
print(f"binary filter n={n} bytes={total/1e9:.2f}GB selected={out.length} out_bytes={ob/1e6:.1f}MB: {ms:.3f} ms = {n/ms/1e6:.1f} Grows/s, {alg/ms/1e6:.1f} GB/s algorithmic", flush=True)
idx = amd.compute.get_take_indices(m)
ms2, out2 = timed(lambda: amd.compute.take(vals, idx, boundscheck=False))
print(f"binary take of {idx.length} monotonic uint32 indices: {ms2:.3f} ms = {idx.length/ms2/1e6:.2f} Grows/s out", flush=True)
perm = torch.randint(0, n, (idx.length,), dtype=torch.int32, device=dev, generator=g)
ri = A.Array(A.int32, idx.length, [None, perm.view(torch.uint8)], 0, 0)
ms3, out3 = timed(lambda: amd.compute.take(vals, ri, boundscheck=False))
print(f"binary take of {idx.length} random int32 indices: {ms3:.3f} ms = {idx.length/ms3/1e6:.2f} Grows/s out", flush=True)
# stages
from arrow_amd import tracing
try:
    import pyarrow as pa, pyarrow.compute as pc
    s = min(n, 1 << 24)
    hv = vals.slice(0, s).to_pyarrow(); hm = m.slice(0, s).to_pyarrow()
    pc.filter(hv, hm); t = time.perf_counter(); r = 3
    for _ in range(r): ref = pc.filter(hv, hm)
    dt = (time.perf_counter() - t) / r
    print(f"pyarrow pc.filter on first {s} rows: {dt*1e3:.2f} ms = {s/dt/1e9:.3f} Grows/s (1 thread)")
    got = amd.compute.filter(vals.slice(0, s), m.slice(0, s)).to_pyarrow()
    print("parity vs pyarrow on the sample:", got.equals(ref))
except Exception as ex:
    print("pyarrow leg skipped:", ex)
