"""GPU experiment: end-to-end group_by_sum (init + consume + export + finalize) kernel breakdown."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
keys = torch.randint(0, groups, (n,), dtype=torch.int32, device=dev, generator=g)
kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
cap = 1
while cap < 2 * groups + 2:
    cap <<= 1
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = amd.compute.GroupBySum(cap, dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    op.consume(kk, vv)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = op.finalize()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"iter {it}: init {1e3*(t1-t0):.2f} ms consume {1e3*(t2-t1):.2f} ms finalize {1e3*(t3-t2):.2f} ms total {1e3*(t3-t0):.2f}  groups {r[0].numel()}", flush=True)
