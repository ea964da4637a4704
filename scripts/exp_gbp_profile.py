"""Phase timestamps of the group-by's flat level (variant library built with -DARX_GBP_PROFILE): one 2^30-row call, then
the sampled workgroups' phase durations in microseconds (s_memrealtime, 100 MHz)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", 1 << 31))
g = torch.Generator(device=dev).manual_seed(1)
keys = torch.empty(n, dtype=torch.int32, device=dev); vals = torch.empty(n, dtype=torch.int64, device=dev)
for b in range(0, n, 1 << 27):
    e = min(n, b + (1 << 27))
    keys[b:e] = torch.randint(0, 10_000_000, (e - b,), dtype=torch.int32, device=dev, generator=g)
    vals[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=dev, generator=g)
kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
for _ in range(2):
    out = amd.compute.group_by_sum(kk, vv, capacity=1 << 25)
    torch.cuda.synchronize()
buf = np.zeros(64 * 8, np.uint64)
raw = ctypes.CDLL(amd._lib.get_lib()._name)
assert raw.arx_debug_gbp_profile(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(64, 8).astype(np.int64)
t = t[t[:, 7] > t[:, 0]]
d = np.diff(t, axis=1) / 100.0
names = ["load+rank", "scan+cursor atomics", "positions", "round1 LDS write", "round1 store", "round2", "round3+tail"]
print(f"{len(t)} sampled workgroups; whole tile {((t[:,7]-t[:,0])/100.0).mean():.1f} us (min {((t[:,7]-t[:,0])/100.0).min():.1f}, max {((t[:,7]-t[:,0])/100.0).max():.1f})")
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {d[:, i].mean():7.2f} us   min {d[:, i].min():7.2f}   max {d[:, i].max():7.2f}")
