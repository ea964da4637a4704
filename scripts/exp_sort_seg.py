"""A/B of the MSD sort's segment fan-out / bucket-kernel variants (same process, interleaved).
Every configuration must produce the same (stable) permutation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
lib = amd._lib.get_lib()
dev = torch.device("cuda", 0)
DEFAULTS = {"sort_msd_seg_min_bits": 1, "sort_msd_final_rows_log2": 4, "sort_msd_small_bucket": 0,
            "sort_msd_segment_rows": 1 << 27}
CONFIGS = [("f4 (old)", {}), ("f3", {"sort_msd_final_rows_log2": 3}), ("f2", {"sort_msd_final_rows_log2": 2}),
           ("f1", {"sort_msd_final_rows_log2": 1}), ("f3+small", {"sort_msd_final_rows_log2": 3, "sort_msd_small_bucket": 1}),
           ("f5", {"sort_msd_final_rows_log2": 5})]
for n in (2_000_000_000, 1 << 30, 1 << 27, 1 << 24):
    g = torch.Generator(device=dev).manual_seed(3)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        k[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=dev, generator=g)
    ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
    def run():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = amd.compute.sort_indices(ak); e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e), out
    run()
    best, ref = {}, None
    for rep in range(3):
        for name, opts in CONFIGS:
            for kk, vv in {**DEFAULTS, **opts}.items():
                assert lib.arx_set_option(kk.encode(), vv) == 0
            ms, out = run()
            best[name] = min(best.get(name, 1e9), ms)
            if rep == 0:
                idx = out.data[: n * 8].view(torch.int64)
                if ref is None:
                    ref = idx.clone()
                else:
                    assert torch.equal(ref, idx), name
            del out
    for name, ms in best.items():
        print(f"n={n} {name:18s} {ms:8.2f} ms  {n / ms / 1e6:6.1f} Grows/s", flush=True)
    del k, ak, ref
