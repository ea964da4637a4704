"""Grouper + dense hash_sum on int64 keys over their whole range (what the fused 32-bit operator cannot take) and on
two int32 key columns: time per stage.  ROWS (default 2^28), GROUPS (default 10M)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402
from arrow_amd.array import int32, int64  # noqa: E402

rows = int(os.environ.get("ROWS", 1 << 28))
groups = int(os.environ.get("GROUPS", 10_000_000))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
pool = torch.randint(-2**63, 2**63 - 1, (groups,), dtype=torch.int64, device=dev, generator=g)
idx = torch.randint(0, groups, (rows,), dtype=torch.int64, device=dev, generator=g)
k64 = pool[idx]
del idx
vals = torch.randint(-2**40, 2**40, (rows,), dtype=torch.int64, device=dev, generator=g)
kk = amd.Array(int64, rows, [None, k64.view(torch.uint8)], 0, 0)
vv = amd.Array(int64, rows, [None, vals.view(torch.uint8)], 0, 0)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, out


def first_batch():
    gr = amd.compute.Grouper([int64], groups + 16)
    return gr, gr.consume([kk])


ms, (gr, ids) = timed(first_batch)
print(f"Grouper.consume int64 keys, {rows} rows, every group new ({gr.num_groups} groups): {ms:8.2f} ms  {rows / ms / 1e6:6.2f} Grows/s")
ms, ids2 = timed(lambda: gr.consume([kk]))
print(f"Grouper.consume again (no new group: probe only):                              {ms:8.2f} ms  {rows / ms / 1e6:6.2f} Grows/s")
assert torch.equal(ids.data[: rows * 4], ids2.data[: rows * 4])
ms, _ = timed(lambda: amd.compute.group_by([kk], [(vv, "hash_sum")], max_groups=groups + 16), reps=2)
print(f"group_by([int64 key], sum) end to end:                                            {ms:8.2f} ms  {rows / ms / 1e6:6.2f} Grows/s")
a = amd.Array(int32, rows, [None, k64.view(torch.int32)[::2].contiguous().view(torch.uint8)], 0, 0)
b = amd.Array(int32, rows, [None, k64.view(torch.int32)[1::2].contiguous().view(torch.uint8)], 0, 0)
ms, _ = timed(lambda: amd.compute.group_by([a, b], [(vv, "hash_sum")], max_groups=groups + 16), reps=2)
print(f"group_by([int32, int32], sum) end to end:                                         {ms:8.2f} ms  {rows / ms / 1e6:6.2f} Grows/s")
print("GROUPER_EXP_OK")
