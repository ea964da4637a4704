import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
for lg in (12, 14, 16, 18, 20, 22):
    n = 1 << lg
    k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
    ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
    row = [f"n=2^{lg}"]
    for msd in (0, 1):
        lib.arx_set_option(b"sort_msd", msd)
        for _ in range(3): amd.compute.sort_indices(ak)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): amd.compute.sort_indices(ak)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        row.append(f"{'MSD' if msd else 'LSD'} {ms*1e3:.0f} us")
    print(" | ".join(row), flush=True)
