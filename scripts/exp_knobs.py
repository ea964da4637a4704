"""End-to-end A/B of arx_set_option knobs on configs[4] (sort 2e9) and configs[3] (group-by 4e9 / 10M keys), one
process, interleaved.  Usage: exp_knobs.py sort|groupby "k=v k=v" "k=v" ...   (each quoted argument = one configuration;
"" = defaults).  Every knob named anywhere is reset to its default (given as k=v in DEFAULTS env) between runs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402

what = sys.argv[1]
configs = sys.argv[2:] or [""]
rows = int(os.environ.get("ROWS", 2_000_000_000 if what == "sort" else 4_000_000_000))
dev = torch.device("cuda", 0)
lib = amd._lib.get_lib()
defaults = dict(kv.split("=") for kv in os.environ.get("DEFAULTS", "").split())


def apply(cfg):
    for k, v in defaults.items():
        assert lib.arx_set_option(k.encode(), int(v)) == 0, k
    for kv in cfg.split():
        k, v = kv.split("=")
        assert k in defaults, f"give the default of {k} in DEFAULTS"
        assert lib.arx_set_option(k.encode(), int(v)) == 0, kv


def fill(t, lo, hi, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for b in range(0, t.numel(), 1 << 27):
        e = min(t.numel(), b + (1 << 27))
        t[b:e] = torch.randint(lo, hi, (e - b,), dtype=t.dtype, device=dev, generator=g)


if what == "sort":
    k = torch.empty(rows, dtype=torch.int64, device=dev)
    rb = int(os.environ.get("RANGE_BITS", 64))     # keys uniform in [0, 2^RANGE_BITS): ids / timestamps share their top bits
    if rb >= 64:
        fill(k, -2**63, 2**63 - 1, 10)
    else:
        fill(k, 0, 2**rb, 10)
    ak = amd.Array(amd.array.uint64, rows, [None, k.view(torch.uint8)], 0, 0)
    run = lambda: amd.compute.sort_indices(ak)  # noqa: E731
    check = lambda out: int(out.data[: rows * 8].view(torch.int64)[:: max(1, rows // 1000)].sum().item())  # noqa: E731
else:
    keys = torch.empty(rows, dtype=torch.int32, device=dev)
    vals = torch.empty(rows, dtype=torch.int64, device=dev)
    fill(keys, 0, int(os.environ.get("GROUPS", 10_000_000)), 8)
    fill(vals, -2**63, 2**63 - 1, 9)
    kk = amd.Array(amd.array.int32, rows, [None, keys.view(torch.uint8)], 0, 0)
    vv = amd.Array(amd.array.int64, rows, [None, vals.view(torch.uint8)], 0, 0)
    run = lambda: amd.compute.group_by_sum(kk, vv, capacity=1 << 25)  # noqa: E731
    check = lambda out: (int(out[0].numel()), int(out[2].sum().item()))  # noqa: E731
apply("")
run()
torch.cuda.synchronize()
best, ref = {}, None
for rep in range(3):
    for cfg in configs:
        apply(cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        best[cfg] = min(best.get(cfg, 1e9), ms)
        print(f"  rep {rep} [{cfg or 'defaults'}] {ms:.2f} ms", file=sys.stderr, flush=True)
        if rep == 0:
            c = check(out)
            ref = c if ref is None else ref
            assert c == ref, (cfg, c, ref)
        del out
for cfg, ms in best.items():
    print(f"{what} {rows} rows [{cfg or 'defaults'}]: {ms:8.2f} ms  {rows / ms / 1e6:6.1f} Grows/s", flush=True)
