import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
n = int(sys.argv[1])
k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
lib.arx_set_option(b"sort_msd", 1)
for _ in range(3):
    amd.compute.sort_indices(ak)
torch.cuda.synchronize()
