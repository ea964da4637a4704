"""One warm + K timed runs of configs[4] (array_sort_indices, 2e9 uint64 rows) and configs[3] (hash_sum, 4e9 rows /
10M int32 keys) on ONE GPU — the command rocprofv3 wraps for the kernel-trace and PMC (FETCH_SIZE / WRITE_SIZE)
summaries of the sort and group-by kernels under profiles/.  Usage: prof_sort_groupby.py [sort|groupby|both] [runs]
[sort_rows] [groupby_rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sort_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000_000
gb_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 4_000_000_000
dev = torch.device("cuda", 0)
lib = amd._lib.get_lib()
for kv in os.environ.get("ARX_OPTIONS", "").split():
    k, v = kv.split("=")
    assert lib.arx_set_option(k.encode(), int(v)) == 0, kv


def fill(t, lo, hi, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for b in range(0, t.numel(), 1 << 27):
        e = min(t.numel(), b + (1 << 27))
        t[b:e] = torch.randint(lo, hi, (e - b,), dtype=t.dtype, device=dev, generator=g)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out


if what == "copy":   # the same counters on a plain streaming copy (arx_buffer_copy, 16 GB moved twice): the baseline
    src = torch.empty(sort_rows * 8, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.view(torch.int64).fill_(7)
    for i in range(runs + 1):
        ms, _ = timed(lambda: amd.compute.copy_buffer(src, dst))
        print(f"copy {src.numel()} bytes run {i}: {ms:.3f} ms = {2 * src.numel() / ms / 1e6:.0f} GB/s", flush=True)
    del src, dst
    torch.cuda.empty_cache()
if what in ("sort", "both"):
    k = torch.empty(sort_rows, dtype=torch.int64, device=dev)
    fill(k, -2**63, 2**63 - 1, 10)
    ak = amd.Array(amd.array.uint64, sort_rows, [None, k.view(torch.uint8)], 0, 0)
    for i in range(runs + 1):
        ms, out = timed(lambda: amd.compute.sort_indices(ak))
        print(f"sort_indices {sort_rows} rows run {i}: {ms:.2f} ms = {sort_rows / ms / 1e6:.1f} Grows/s", flush=True)
        del out
    del k, ak
    torch.cuda.empty_cache()
if what in ("groupby", "both"):
    keys = torch.empty(gb_rows, dtype=torch.int32, device=dev)
    vals = torch.empty(gb_rows, dtype=torch.int64, device=dev)
    fill(keys, 0, 10_000_000, 8)
    fill(vals, -2**63, 2**63 - 1, 9)
    kk = amd.Array(amd.array.int32, gb_rows, [None, keys.view(torch.uint8)], 0, 0)
    vv = amd.Array(amd.array.int64, gb_rows, [None, vals.view(torch.uint8)], 0, 0)
    for i in range(runs + 1):
        ms, out = timed(lambda: amd.compute.group_by_sum(kk, vv, capacity=1 << 25))
        print(f"group_by_sum {gb_rows} rows run {i}: {ms:.2f} ms = {gb_rows / ms / 1e6:.1f} Grows/s, {out[0].numel()} groups",
              flush=True)
        del out
