"""GPU experiment: sort_indices on skewed key distributions (equal-width MSD buckets overflow):
LSD passes vs the sampled-splitter MSD form."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(9)
n = 1 << 27
x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
cases = {
    "float64 N(0,1)": (amd.array.float64, x),
    "float32 N(0,1)": (amd.array.float32, x.to(torch.float32)),
    "int32 N(0,1e9)": (amd.array.int32, (x * 1e9).clamp(-2e9, 2e9).to(torch.int32)),
    "int64 timestamps (clustered)": (amd.array.int64, (1_700_000_000_000_000 + (x.abs() * 3.6e9)).to(torch.int64)),
    "uint64 uniform": (amd.array.uint64, torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)),
}
for name, (t, arr) in cases.items():
    a = amd.Array(t, n, [None, arr.contiguous().view(torch.uint8)], 0, 0)
    row = [name]
    for label, msd, sampled in (("LSD", 0, 0), ("MSD auto", -1, 1)):
        lib.arx_set_option(b"sort_msd", msd); lib.arx_set_option(b"sort_msd_sampled", sampled)
        amd.compute.sort_indices(a); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); amd.compute.sort_indices(a); amd.compute.sort_indices(a); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 2
        row.append(f"{label} {ms:.3f} ms ({n / ms / 1e6:.1f} Grows/s)")
    print(" | ".join(row), flush=True)
