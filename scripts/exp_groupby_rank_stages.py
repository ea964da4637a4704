"""Per-stage times of ONE rank of the 8-way sharded hash_sum (4B rows / 10M keys => 500M rows per
rank): local consume, export, partition of partials, merge of the received partials (emulated with
this rank's own partials: same count, 1/8 of the key space per source), finalize.  The exchange
itself needs 8 GPUs; everything else is what bounds the scaling."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import parallel
from arrow_amd.compute import GroupBySum
dev = torch.device("cuda", 0)
world = int(os.environ.get("WORLD", 8))
n = 4_000_000_000 // world
groups = 10_000_000
g = torch.Generator(device=dev).manual_seed(1)
keys = torch.empty(n, dtype=torch.int32, device=dev); vals = torch.empty(n, dtype=torch.int64, device=dev)
for b in range(0, n, 1 << 26):
    e = min(n, b + (1 << 26))
    keys[b:e] = torch.randint(0, groups, (e - b,), dtype=torch.int32, device=dev, generator=g)
    vals[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=dev, generator=g)
kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
cap = 1
while cap < 2 * groups + 2: cap <<= 1
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(3):
    t = [ev() for _ in range(7)]
    t[0].record()
    local = GroupBySum(cap, dev); t[1].record()
    local.consume(kk, vv); t[2].record()
    t[3].record()
    records, counts = parallel.export_partitioned(local, world); t[4].record()
    c = counts.cpu().tolist()
    # emulate the received partials: `world` blocks of the size this rank would receive
    mine = records[: c[0] * parallel.RECORD_BYTES]
    owned = GroupBySum(max(16, 2 * world * c[0] + 2), dev)
    for r in range(world):
        parallel.merge_records(owned, mine)
    t[5].record()
    out = owned.finalize(); t[6].record()
    torch.cuda.synchronize()
    names = ["init", "consume", "-", "export_partitioned", "merge(8x)+init", "finalize"]
    print(" | ".join(f"{nm} {t[i].elapsed_time(t[i+1]):.2f}" for i, nm in enumerate(names)),
          "| total %.2f ms" % t[0].elapsed_time(t[6]), "| groups/rank", c[0], flush=True)
