"""A/B of the group-by consume knobs on one box (same process, interleaved): rows x keys grid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd.compute import GroupBySum
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 1 << 29))
groups = 10_000_000
g = torch.Generator(device=dev).manual_seed(1)
keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
cap = 1 << 25
want = int(vals.sum().item())
def run():
    st = GroupBySum(cap, dev)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); st.consume(kk, vv); e.record(); torch.cuda.synchronize()
    ks, kv, sums, valid = st.finalize()
    assert int(sums.sum().item()) == want and sums.numel() == groups, (int(sums.numel()))
    return s.elapsed_time(e)
configs = [("bits auto(13)", {"groupby_partition_bits": -1}), ("bits=12", {"groupby_partition_bits": 12}),
           ("bits=14", {"groupby_partition_bits": 14}), ("bits=12,b1=6", {"groupby_partition_bits": 12, "groupby_b1": 6})]
run(); run()
best = {}
for rep in range(4):
    for name, opts in configs:
        for k, v in opts.items():
            assert lib.arx_set_option(k.encode(), v) == 0
        ms = run()
        best[name] = min(best.get(name, 1e9), ms)
for name, _ in configs:
    print(f"{name:12s} {best[name]:.3f} ms  {n / best[name] / 1e6:.1f} Grows/s", flush=True)
