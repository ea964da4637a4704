"""Throughput of the HashAggregateKernel consume (hash_sum(int64, uint32 group id)): per-row device atomics vs the
scratch form (partition by group id + LDS aggregation), 2^28 rows, from 100 hot groups to 10M groups."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402

lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
n = 1 << 28
g = torch.Generator(device=dev).manual_seed(5)
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
stream = int(torch.cuda.current_stream(dev).cuda_stream)
for groups in (100, 10_000, 1_000_000, 10_000_000):
    ids = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
    span = amd._lib.ArxSpan(None, vals.data_ptr(), 0, n, 0)
    want = torch.zeros(groups, dtype=torch.int64, device=dev).index_add_(0, ids.to(torch.int64), vals)
    for form in ("atomics", "partitioned"):
        sums = torch.zeros(groups, dtype=torch.int64, device=dev)
        counts = torch.zeros(groups, dtype=torch.int64, device=dev)
        nulls = torch.zeros(groups, dtype=torch.int32, device=dev)
        ws_bytes = lib.arx_hash_sum_consume_workspace_bytes(n, groups) if form == "partitioned" else 0
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) & ~255

        def run():
            amd._lib.check(lib.arx_hash_sum_i64_consume_ws(C.byref(span), 0, 0, ids.data_ptr(), n, groups, sums.data_ptr(),
                                                           counts.data_ptr(), nulls.data_ptr(),
                                                           ws_ptr if ws_bytes else None, ws_bytes, stream))
        run()
        torch.cuda.synchronize()
        assert torch.equal(sums, want) and int(counts.sum().item()) == n, (groups, form)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 3
        print(f"hash_sum vtable consume, {n} rows, {groups:>9d} groups, {form:12s}: {ms:8.2f} ms = {n / ms / 1e6:6.1f} Grows/s "
              f"({12 * n / ms / 1e6:7.1f} GB/s algorithmic)", flush=True)
