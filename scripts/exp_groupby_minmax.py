"""hash_min/hash_max consume rate on the fused table (direct HBM atomics), rows x groups grid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd.compute import GroupBySum
dev = torch.device("cuda", 0)
n = 1 << 28
g = torch.Generator(device=dev).manual_seed(1)
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
for groups in (100, 10_000, 1_000_000, 10_000_000):
    keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
    kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    cap = 16
    while cap < 2 * groups + 2: cap <<= 1
    best = 1e9
    for rep in range(3):
        st = GroupBySum(cap, dev)
        st.consume(kk, vv)                    # the table already holds the keys (the usual case next to a sum)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); st.consume_min_max(kk, vv); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    ks, kv, mn, mx, ok = st.finalize_min_max()
    assert int(mn.min()) == int(vals.min()) and int(mx.max()) == int(vals.max()) and mn.numel() == groups
    print(f"groups {groups:>9}: min/max consume of 2^28 rows {best:.2f} ms = {n / best / 1e6:.1f} Grows/s", flush=True)
