"""GPU experiment: throughput of the global-atomics group-by consume vs number of groups
(is an L2/MALL-resident table fast enough to skip LDS partitioning?)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n = 1 << 28
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
out = {}
for groups in [100, 1000, 10_000, 100_000, 1_000_000, 10_000_000]:
    keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
    kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    cap = 1
    while cap < 2 * groups + 2:
        cap <<= 1
    op = amd.compute.GroupBySum(cap, dev)
    op.consume(kk, vv)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        op.consume(kk, vv)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    out[groups] = {"ms": round(ms, 3), "grows_per_s": round(n / ms / 1e6, 2), "table_MB": round(cap * 28 / 1e6, 1)}
    print(groups, out[groups], flush=True)
print(json.dumps(out))
