"""One virtual rank of the sharded hash_sum at P = 8 (5e8 rows / 1e7 keys) — the local pass + merge + finalize, four times,
for a rocprofv3 kernel trace: which kernels (and which gaps between them) make up the ~1.5 ms a rank spends above
its 1/8 share of the one-GPU time.  DIRECT=1: the local pass without the local table (consume_partials)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import parallel
from arrow_amd.array import Array
from arrow_amd.compute import GroupBySum
dev = torch.device("cuda", 0)
world, direct = int(os.environ.get("WORLD", 8)), os.environ.get("DIRECT", "1") == "1"
n, groups, cap = 4_000_000_000 // world, 10_000_000, 1 << 25
g = torch.Generator(device=dev).manual_seed(8)
keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
kk = Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0); vv = Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    local = None if direct else GroupBySum(cap, dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if direct:
        records, counts = parallel.consume_partials(kk, vv, cap, world)
    else:
        local.consume(kk, vv)
        torch.cuda.synchronize(); t1b = time.perf_counter()
        records, counts = parallel.export_partitioned(local, world)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    c = counts.cpu().tolist()
    mine = records[: c[0] * parallel.RECORD_BYTES]
    owned = GroupBySum(max(16, int(float(os.environ.get('OWNED_FACTOR', 2 * world)) * c[0]) + 2), dev)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    for r in range(world):
        parallel.merge_records(owned, mine)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    out = owned.finalize()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"rep {rep}: state {1e3 * (t1 - t0):.2f}  local pass {1e3 * (t2 - t1):.2f}  owned state {1e3 * (t3 - t2):.2f}  merge {1e3 * (t4 - t3):.2f}  finalize {1e3 * (t5 - t4):.2f}"
          f"  records {sum(c)}", flush=True)
    del local, records, owned, out, mine
