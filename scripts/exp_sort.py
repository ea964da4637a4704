import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
for n in (1 << 24, 1 << 27, 1 << 30):
    k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
    ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
    for msd in (0, 1, 0, 1):
        lib.arx_set_option(b"sort_msd", msd)
        amd.compute.sort_indices(ak); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); amd.compute.sort_indices(ak); amd.compute.sort_indices(ak); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)/2
        print("n", n, "msd", msd, round(ms,3), "ms", round(n/ms/1e6,2), "Grows/s", flush=True)
    del k, ak
